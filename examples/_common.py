"""shared by the example applications: locating the model arrays and writing a synthetic MNIST-format file"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def synthetic_mnist_file(path, records, seed=1):
    """`DataPreprocess/GetMNIST.cs` output format (`label <TAB> 784 <TAB> index:value ...`) with MNIST-like sparsity (a pixel is
    0 with probability 0.81) and random labels - for timing and plumbing when MNIST-28x28-test.txt is not at hand."""
    r = np.random.default_rng(seed)
    with open(path, "w") as f:
        for _ in range(records):
            img = np.where(r.random(784) < 0.81, 0, r.integers(1, 256, size=784))
            nz = np.nonzero(img)[0]
            f.write("%d\t784\t%s\n" % (int(r.integers(0, 10)), "\t".join("%d:%d" % (i, int(img[i])) for i in nz)))
    return path
