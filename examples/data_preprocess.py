"""The reference's `DataPreprocess` program for the two data sets the BASELINE networks read (`DataPreprocess/GetMNIST.cs`,
`GetCIFAR.cs`): the raw distribution files -> the TSV files the reader layers parse.

    python examples/data_preprocess.py mnist [dir]     # t10k-images-idx3-ubyte.gz + t10k-labels-idx1-ubyte.gz -> MNIST-28x28-test.txt
    python examples/data_preprocess.py cifar [dir]     # cifar-10-binary.tar.gz -> cifar-test.tsv

MNIST records are sparse (`label <TAB> 784 <TAB> index:value ...`, zero pixels left out, GetMNIST.cs:37-46); CIFAR records are dense
(`label <TAB> 3072 values`), and GetCIFAR writes value (colour, y, x) from byte y + 32*(x + 32*colour) of the record - i.e. each
colour plane transposed (GetCIFAR.cs:24-27); the trained CIFAR model expects exactly that order.  (The Caltech-101 features need an
ML.NET AlexNet featuriser, GetCAL.cs: not covered.)
"""
import gzip
import os
import sys
import tarfile

import numpy as np


def mnist(directory="."):
    img_path, lab_path = (os.path.join(directory, f) for f in ("t10k-images-idx3-ubyte.gz", "t10k-labels-idx1-ubyte.gz"))
    if not (os.path.exists(img_path) and os.path.exists(lab_path)):
        print("Please download the following files from http://yann.lecun.com/exdb/mnist/\n\tt10k-images-idx3-ubyte.gz\n\tt10k-labels-idx1-ubyte.gz")
        return None
    print("reading input files")
    images_bin, labels_bin = gzip.open(img_path, "rb").read(), gzip.open(lab_path, "rb").read()
    if labels_bin[:4] != bytes([0, 0, 8, 1]):
        raise Exception("labels file magic number currepted")
    if images_bin[:4] != bytes([0, 0, 8, 3]):
        raise Exception("images file magic number currepted")
    labels = np.frombuffer(labels_bin, dtype=np.uint8, offset=8)
    images = np.frombuffer(images_bin, dtype=np.uint8, offset=16, count=784 * labels.size).reshape(labels.size, 784)
    out = os.path.join(directory, "MNIST-28x28-test.txt")
    print("writing MNIST-28x28-test.txt")
    with open(out, "w") as f:
        for lab, img in zip(labels, images):
            nz = np.nonzero(img)[0]
            f.write("%d\t%d%s\n" % (lab, 784, "".join("\t%d:%d" % (j, img[j]) for j in nz)))
    print("done")
    return out


def cifar(directory="."):
    tar_path = os.path.join(directory, "cifar-10-binary.tar.gz")
    if not os.path.exists(tar_path):
        print("Please download the binary version of the CIFAR-10 dataset from https://www.cs.toronto.edu/~kriz/cifar-10-binary.tar.gz")
        return None
    print("reading cifar-10-binary.tar.gz")
    with tarfile.open(tar_path, "r:gz") as tar:
        member = next(m for m in tar.getmembers() if m.name.replace("\\", "/").endswith("cifar-10-batches-bin/test_batch.bin"))
        print("reading test_batch.bin")
        raw = np.frombuffer(tar.extractfile(member).read(), dtype=np.uint8)
    rec = raw.reshape(-1, 3 * 32 * 32 + 1)                        # the +1 is the label column
    planes = rec[:, 1:].reshape(-1, 3, 32, 32)                       # [record][colour][x][y] in GetCIFAR's naming: byte y + 32*(x + 32*colour)
    values = planes.transpose(0, 1, 3, 2).reshape(rec.shape[0], -1)  # written colour, y, x
    out = os.path.join(directory, "cifar-test.tsv")
    print("writing cifar-test.tsv")
    with open(out, "w") as f:
        for lab, row in zip(rec[:, 0], values):
            f.write("%d\t%s\n" % (lab, "\t".join(map(str, row))))
    print("done")
    return out


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in ("mnist", "cifar"):
        print(__doc__)
        sys.exit(2)
    {"mnist": mnist, "cifar": cifar}[sys.argv[1]](*(sys.argv[2:3]))
