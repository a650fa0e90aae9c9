"""The reference's CryptoNets application (`CryptoNets/CryptoNets.cs`): MNIST in batches of 8192 images (one per slot) through
conv -> square -> dense -> square -> dense under BFV, N = 8192, plaintext primes {549764251649, 549764284417}, on the MI355X.

    python examples/cryptonets.py --file MNIST-28x28-test.txt                # the reference's run: 10000 records, accuracy printed
    python examples/cryptonets.py --synthetic 8192                           # no data set at hand: one synthetic batch (timing)
    python examples/cryptonets.py --synthetic 8192 --raw                     # the plaintext factory instead (debugging)
"""
import argparse
import tempfile
import time

import numpy as np

from _common import GOLDEN, synthetic_mnist_file
from cryptonets_amd import networks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--file", default="MNIST-28x28-test.txt")
    ap.add_argument("--synthetic", type=int, default=0, metavar="RECORDS")
    ap.add_argument("--records", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--raw", action="store_true", help="RawFactory (plaintext) instead of EncryptedSealBfvFactory")
    ap.add_argument("--weights", default=GOLDEN + "/cryptonets_weights.npz")
    a = ap.parse_args()
    if a.synthetic:
        a.file, a.records = synthetic_mnist_file(tempfile.mktemp(suffix=".tsv"), a.synthetic), a.synthetic
    start = time.time()
    if a.raw:
        from cryptonets_amd.raw import RawFactory
        Factory = RawFactory(a.batch)
    else:
        from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
        Factory = EncryptedSealBfvFactory(networks.FACTORY_PARAMETERS["CryptoNets"]["primes"], a.batch, galois=False)
    print("Generated keys in %.2f seconds" % (time.time() - start))
    reader = networks.mnist_reader(a.file, a.batch)
    network = networks.CryptoNets(Factory, reader, np.load(a.weights))
    print("Preparing")
    errs, count = networks.evaluate_batches(network, Factory, reader, a.records)
    print("errs %d/%d accuracy %.3f%%" % (errs, count, 100 - 100.0 * errs / max(count, 1)))


if __name__ == "__main__":
    main()
