"""The reference's `Basic Example/Program.cs` (BASELINE config 0): squared norm, sum of the elements and elementwise product of
two 3-element vectors through the IVector surface.

    python examples/basic_example.py          # EncryptedSealBfvFactory(): N = 4096, plaintext primes {40961 ... 188417}, on the MI355X
    python examples/basic_example.py --raw    # RawFactory(4096): the plaintext path (Program.cs:16), no GPU
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cryptonets_amd.hewrapper import EVectorFormat        # noqa: E402
from cryptonets_amd.utils import ProcessInEnv             # noqa: E402


def run(Factory):
    v = np.array([1.0, 2.0, 3.0])
    z = np.array([-1.0, 5.0, -4.0])
    out = {}

    def compute(env):                                        # Program.cs:35-47
        ciphertext = Factory.GetEncryptedVector(v, EVectorFormat.dense, 1)
        out["norm"] = ciphertext.DotProduct(ciphertext, env).Decrypt(env)
        out["total"] = ciphertext.SumAllSlots(env).Decrypt(env)
        z_ciphertext = Factory.GetEncryptedVector(z, EVectorFormat.dense, 1)
        out["prod"] = ciphertext.PointwiseMultiply(z_ciphertext, env).Decrypt(env)
    ProcessInEnv(compute, Factory)
    norm, total, prod = out["norm"], out["total"], out["prod"]
    return {"norm_squared": [float(x) for x in norm], "sum": [float(x) for x in total], "elementwise": [float(x) for x in prod]}


def main(argv):
    start = time.time()
    if "--raw" in argv:
        from cryptonets_amd.raw import RawFactory
        Factory = RawFactory(4096)
    else:
        from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
        Factory = EncryptedSealBfvFactory()                  # keys are generated here (Program.cs:18)
    print("Generated keys in %.2f seconds" % (time.time() - start))
    start = time.time()
    out = run(Factory)
    print("Norm Sqared is:\n%s" % out["norm_squared"])
    print("sum of elements in a vector:\n%s" % out["sum"])
    print("elementwise multiply = \n%s" % out["elementwise"])
    print("Compute in %.2f seconds" % (time.time() - start))


if __name__ == "__main__":
    main(sys.argv[1:])
