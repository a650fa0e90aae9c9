"""The example applications (`examples/`: the reference's `Basic Example`, `CryptoNets` and `LowLatencyCryptoNets` mains) run as
programs.  CPU: on the plaintext factory.  GPU: encrypted on the MI355X, and the predictions must be the ones the plaintext factory
makes on the same records (the reference's own cross-check: the `-e` switch, `LoLaCryptonets.cs:19-20`)."""
import os
import re
import subprocess
import sys

import pytest

EX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples")


def run(script, *args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(EX, script)] + list(args), capture_output=True, text=True, timeout=timeout, cwd=EX)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def predictions(out):
    return [(int(a), int(b)) for a, b in re.findall(r"prediction (\d+) label (\d+)", out)]


def test_basic_example_program_raw():
    out = run("basic_example.py", "--raw")
    assert "[14.0]" in out and "[6.0]" in out and "[-1.0, 10.0, -12.0]" in out


def test_cryptonets_program_raw():
    out = run("cryptonets.py", "--synthetic", "48", "--batch", "32", "--raw")
    assert re.search(r"errs \d+/48 accuracy", out) and out.count("Batch size") == 2          # 32 + 16 records


@pytest.mark.parametrize("net", ["LoLa", "LoLaDense", "LoLaSmall", "LoLaLarge"])
def test_lola_program_raw(net):
    out = run("lola.py", "-n", net, "--synthetic", "3", "-v")
    assert len(predictions(out)) == 3 and "Maximal value used" in out and "Layer LLPoolLayer computed in" in out


@pytest.mark.gpu
def test_basic_example_program_encrypted():
    out = run("basic_example.py")
    assert "[14.0]" in out and "[6.0]" in out and "[-1.0, 10.0, -12.0" in out


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["LoLa", "LoLaSmall", "LoLaDense", "LoLaLarge"])
def test_lola_program_encrypted_predicts_like_raw(net):
    n = "2" if net in ("LoLaDense", "LoLaLarge") else "4"
    enc = predictions(run("lola.py", "-n", net, "-e", "--synthetic", n))
    raw = predictions(run("lola.py", "-n", net, "--synthetic", n))
    assert len(enc) == int(n) and enc == raw


@pytest.mark.gpu
def test_cryptonets_program_encrypted_scores_like_raw():
    enc = run("cryptonets.py", "--synthetic", "300")
    raw = run("cryptonets.py", "--synthetic", "300", "--raw")
    pick = lambda s: re.findall(r"errs (\d+)/300 accuracy", s)[-1]
    assert pick(enc) == pick(raw)


@pytest.mark.gpu
def test_lola_program_recorded_evaluation_predicts_like_raw():
    """`--graph`: the first record rehearses, the rest replay the recorded evaluation - same predictions as the plaintext factory"""
    enc = run("lola.py", "-n", "LoLa", "-e", "--graph", "--synthetic", "4")
    raw = run("lola.py", "-n", "LoLa", "--synthetic", "4")
    assert len(predictions(enc)) == 4 and predictions(enc) == predictions(raw) and enc.count("(recorded)") == 3


def test_lola_cifar_program_raw():
    out = run("lola_cifar.py", "--synthetic", "1")
    assert len(predictions(out)) == 1 and "Inference-Time" in out and "Max computed value 2^" in out


@pytest.mark.gpu
def test_lola_cifar_program_encrypted_predicts_like_raw():
    """synthetic model of the reference's shapes (CifarWeight.csv is a missing blob): the encrypted prediction is the plaintext one"""
    enc = predictions(run("lola_cifar.py", "-e", "--limbs", "9", "--synthetic", "1"))
    raw = predictions(run("lola_cifar.py", "--synthetic", "1"))
    assert len(enc) == 1 and enc == raw


@pytest.mark.gpu
def test_verbose_run_reports_operations_and_budget():
    """-v prints the per-layer OperationsCount (BaseLayer.cs:39), --budget the CryptoTracker watermark (BaseLayer.cs:37)"""
    out = run("lola.py", "-n", "LoLa", "-e", "-v", "--budget", "--synthetic", "1")
    assert "Operations:" in out and re.search(r"\tRotation\t[1-9]", out) and re.search(r"\tRelinarization\t[1-9]", out)
    assert re.search(r"Minimal noise budget seen [1-9]\d* bits", out) and "Warning: Current minimal budget" in out


def test_data_preprocess_formats(tmp_path):
    """`DataPreprocess/GetMNIST.cs:37-81`, `GetCIFAR.cs:16-29`: the written records, read back by the reader layers"""
    import gzip
    import importlib.util
    import io
    import tarfile
    import numpy as np
    from cryptonets_amd.layers import BatchReader
    from cryptonets_amd import networks
    spec = importlib.util.spec_from_file_location("data_preprocess", os.path.join(EX, "data_preprocess.py"))
    dp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dp)
    r = np.random.default_rng(2)
    assert dp.mnist(str(tmp_path)) is None and dp.cifar(str(tmp_path)) is None        # files absent: a hint, nothing written
    imgs = np.where(r.random((5, 784)) < 0.8, 0, r.integers(1, 256, size=(5, 784))).astype(np.uint8)
    labs = r.integers(0, 10, size=5).astype(np.uint8)
    with gzip.open(tmp_path / "t10k-images-idx3-ubyte.gz", "wb") as f:
        f.write(bytes([0, 0, 8, 3]) + (5).to_bytes(4, "big") + (28).to_bytes(4, "big") * 2 + imgs.tobytes())
    with gzip.open(tmp_path / "t10k-labels-idx1-ubyte.gz", "wb") as f:
        f.write(bytes([0, 0, 8, 1]) + (5).to_bytes(4, "big") + labs.tobytes())
    path = dp.mnist(str(tmp_path))
    first = open(path).readline().rstrip("\n").split("\t")
    assert first[0] == str(labs[0]) and first[1] == "784" and len(first) == 2 + np.count_nonzero(imgs[0])
    rd = BatchReader(FileName=path, SparseFormat=True, MaxSlots=8)
    m = rd.GetNext()
    assert list(rd.Labels) == list(labs) and np.array_equal(m.Data, imgs.astype(float))
    rec = np.concatenate([r.integers(0, 10, size=(3, 1)), r.integers(0, 256, size=(3, 3072))], axis=1).astype(np.uint8)
    with tarfile.open(tmp_path / "cifar-10-binary.tar.gz", "w:gz") as tar:
        info = tarfile.TarInfo("cifar-10-batches-bin/test_batch.bin")
        info.size = rec.size
        tar.addfile(info, io.BytesIO(rec.tobytes()))
    path = dp.cifar(str(tmp_path))
    rows = [ln.rstrip("\n").split("\t") for ln in open(path)]
    assert len(rows) == 3 and all(len(x) == 3073 for x in rows)
    for i in range(3):
        assert int(rows[i][0]) == rec[i, 0]
        for (c, y, x) in [(0, 0, 0), (1, 5, 9), (2, 31, 30), (0, 17, 3)]:
            assert int(rows[i][1 + (c * 32 + y) * 32 + x]) == rec[i, 1 + y + 32 * (x + 32 * c)]        # GetCIFAR.cs:24-27
    reader = networks.cifar_reader(path)
    m = reader.GetNext()
    assert list(reader.Labels) == [rec[0, 0]] and m.RowCount == 196 and m.ColumnCount == 192
