"""The reference's CIFAR-10 application (`CifarCryptoNet/LolaCifarCryptoNet.cs`): one 3x32x32 image per prediction, N = 16384, eight
coefficient primes, plaintext primes {957181001729, 957181034497}.

    python examples/lola_cifar.py -e --file cifar-test.tsv --weights CifarWeight.csv --biases CifarBias.csv
    python examples/lola_cifar.py -e --synthetic 2          # random model of the same shapes, random records (timing)

The trained CifarWeight.csv is not part of the reference repository (`.MISSING_LARGE_BLOBS`); without it only the synthetic run is
possible.  `--limbs 9` takes the whole CoeffModulus128(16384): full-range synthetic plaintexts need it (LoLa-CIFAR).
"""
import argparse
import tempfile
import time

import numpy as np

from _common import ROOT  # noqa: F401  (puts the repository on sys.path)
from cryptonets_amd import networks
from cryptonets_amd.layers import WeightsReader


def synthetic_model(seed=5):
    r = np.random.default_rng(seed)
    q = lambda a, s: np.rint(a * s) / s
    W = [q(r.normal(0, 0.05, 83 * 192), 256), q(r.normal(0, 0.02, 112 * 8300), 512), q(r.normal(0, 0.05, 10 * 5488), 512)]
    B = [q(r.normal(0, 0.05, 83), 256), q(r.normal(0, 0.05, 112), 512), q(r.normal(0, 0.05, 10), 512)]
    return W, B


def synthetic_cifar_file(path, records, seed=1):
    r = np.random.default_rng(seed)
    with open(path, "w") as f:
        for _ in range(records):
            f.write("%d\t%s\n" % (int(r.integers(0, 10)), "\t".join(str(int(v)) for v in r.integers(0, 256, size=3072))))
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-e", "--encrypt", action="store_true")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--budget", action="store_true", help="with -e -v: probe the invariant noise budget after every layer (CryptoTracker)")
    ap.add_argument("--file", default="cifar-test.tsv")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--biases", default=None)
    ap.add_argument("--synthetic", type=int, default=0, metavar="RECORDS")
    ap.add_argument("--records", type=int, default=10000)
    ap.add_argument("--limbs", type=int, default=None)
    a = ap.parse_args()
    if a.weights:
        wr = WeightsReader(a.weights, a.biases)
        W, B = wr.Weights, wr.Biases
    else:
        W, B = synthetic_model()
    if a.synthetic:
        a.file, a.records = synthetic_cifar_file(tempfile.mktemp(suffix=".tsv"), a.synthetic), a.synthetic
    parms = dict(networks.FACTORY_PARAMETERS["LoLaCifar"])
    print("Generating encryption keys %s" % time.strftime("%X"))
    if a.encrypt:
        from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
        if a.limbs:
            parms["SmallModulusCount"] = a.limbs
        Factory = EncryptedSealBfvFactory(**parms)
    else:
        from cryptonets_amd.raw import RawFactory
        Factory = RawFactory(16 * 1024)
    print("Encryption keys ready %s" % time.strftime("%X"))
    reader = networks.cifar_reader(a.file)
    network = networks.LoLaCifar(Factory, reader, W, B)
    print("Preparing")
    if a.budget:
        from cryptonets_amd.cryptotracker import CryptoTracker
        CryptoTracker.EnableBudgetTests()
    errs, count = networks.evaluate_single(network, Factory, a.records, verbose=a.verbose)
    print("errs %d/%d accuracy %.3f%%" % (errs, count, 100 - 100.0 * errs / max(count, 1)))
    if a.budget and a.encrypt:
        print("Minimal noise budget seen %d bits" % CryptoTracker.MinBudgetSoFar)
    if not a.encrypt:
        from cryptonets_amd.raw import RawMatrix
        print("Max computed value 2^%.2f" % np.log2(RawMatrix.Max))


if __name__ == "__main__":
    main()
