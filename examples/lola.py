"""The reference's low-latency application (`LowLatencyCryptoNets/LoLaCryptonets.cs`): one image per prediction.

    python examples/lola.py -n LoLa -e --file MNIST-28x28-test.txt          # as the reference: -n network, -e encrypt, -v verbose
    python examples/lola.py -n LoLaSmall -e --synthetic 20                   # synthetic records (timing; labels are random)
    python examples/lola.py -n LoLaDense --synthetic 5                       # without -e: RawFactory, prints the largest value used

Networks: LoLa, LoLaDense, LoLaSmall, LoLaLarge (its MnistLargeWeight.csv is not shipped by the reference: pass --weights /
--biases, or a random model of the same shapes is used).  The coefficient
modulus of LoLaDense / LoLaSmall / LoLaLarge is one prime longer than the reference's (`--limbs`): with the reference's count the noise budget
is exhausted before the last layer (LoLa sections).
"""
import argparse
import math
import tempfile
import time

import numpy as np

from _common import GOLDEN, synthetic_mnist_file
from cryptonets_amd import networks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", "--network", required=True, choices=sorted(networks.LOLA_NETWORKS) + ["LoLaLarge"])
    ap.add_argument("--weights", default=None, help="LoLaLarge: MnistLargeWeight.csv (not shipped by the reference; default: a random model of its shapes)")
    ap.add_argument("--biases", default=None, help="LoLaLarge: MnistLargeBias.csv")
    ap.add_argument("-e", "--encrypt", action="store_true")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--graph", action="store_true", help="with -e: record the evaluation once (one HIP graph per plaintext prime) and replay it for every further record")
    ap.add_argument("--budget", action="store_true", help="with -e -v: probe the invariant noise budget after every layer (CryptoTracker)")
    ap.add_argument("--file", default="MNIST-28x28-test.txt")
    ap.add_argument("--synthetic", type=int, default=0, metavar="RECORDS")
    ap.add_argument("--records", type=int, default=10000)
    ap.add_argument("--limbs", type=int, default=None, help="coefficient primes to take (default: the reference's count + 1 where it needs it)")
    a = ap.parse_args()
    if a.synthetic:
        a.file, a.records = synthetic_mnist_file(tempfile.mktemp(suffix=".tsv"), a.synthetic), a.synthetic
    parms = dict(networks.FACTORY_PARAMETERS[a.network])
    print({"LoLa": "LoLa mode", "LoLaDense": "LoLa-Dense mode", "LoLaSmall": "Small LoLa mode", "LoLaLarge": "Large LoLa mode"}[a.network])
    start = time.time()
    if a.encrypt:
        from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
        if "SmallModulusCount" in parms:
            parms["SmallModulusCount"] = a.limbs if a.limbs else parms["SmallModulusCount"] + 1
        Factory = EncryptedSealBfvFactory(**parms)
    else:
        from cryptonets_amd.raw import RawFactory
        Factory = RawFactory(parms["n"])
    print("Generating keys in %.2f seconds" % (time.time() - start))
    reader = networks.lola_reader(a.network, a.file)
    if a.network == "LoLaLarge":
        if a.weights:
            from cryptonets_amd.layers import WeightsReader
            wr = WeightsReader(a.weights, a.biases)
            W, B = wr.Weights, wr.Biases
        else:
            r = np.random.default_rng(7)                         # small and sparse, so that the logits stay below the 93-bit plaintext modulus
            pick = lambda n, p, s: r.choice([-1.0, 0.0, 1.0], size=n, p=[p / 2, 1 - p, p / 2]) / s
            W = [np.rint(r.normal(0, 0.01, 83 * 64) * 4096) / 16, pick(163 * 83 * 36, 0.01, 64), pick(10 * 2608, 0.02, 512)]
            B = [np.rint(r.normal(0, 0.05, 83) * 4096) / 4096, pick(163, 0.5, 64), pick(10, 0.5, 512)]
        network = networks.LargeLoLa(Factory, reader, W, B)
    else:
        weights = np.load(GOLDEN + ("/small_model_weights.npz" if a.network == "LoLaSmall" else "/cryptonets_weights.npz"))
        network = networks.LOLA_NETWORKS[a.network](Factory, reader, weights)
    if a.budget:
        from cryptonets_amd.cryptotracker import CryptoTracker
        CryptoTracker.EnableBudgetTests()
    if a.graph and a.encrypt:
        errs, count = networks.evaluate_single_recorded(network, Factory, a.records)
    else:
        errs, count = networks.evaluate_single(network, Factory, a.records, verbose=a.verbose)
    print("errs %d/%d accuracy %.3f%%" % (errs, count, 100 - 100.0 * errs / max(count, 1)))
    if a.budget and a.encrypt:
        print("Minimal noise budget seen %d bits" % CryptoTracker.MinBudgetSoFar)
    if a.verbose and not a.encrypt:
        from cryptonets_amd.raw import RawMatrix
        print("Maximal value used %s (%.2f bits)" % (RawMatrix.Max, math.log2(RawMatrix.Max)))


if __name__ == "__main__":
    main()
