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


@pytest.mark.parametrize("net", ["LoLa", "LoLaDense", "LoLaSmall"])
def test_lola_program_raw(net):
    out = run("lola.py", "-n", net, "--synthetic", "3", "-v")
    assert len(predictions(out)) == 3 and "Maximal value used" in out and "Layer LLPoolLayer computed in" in out


@pytest.mark.gpu
def test_basic_example_program_encrypted():
    out = run("basic_example.py")
    assert "[14.0]" in out and "[6.0]" in out and "[-1.0, 10.0, -12.0" in out


@pytest.mark.gpu
@pytest.mark.parametrize("net", ["LoLa", "LoLaSmall", "LoLaDense"])
def test_lola_program_encrypted_predicts_like_raw(net):
    n = "2" if net == "LoLaDense" else "4"
    enc = predictions(run("lola.py", "-n", net, "-e", "--synthetic", n))
    raw = predictions(run("lola.py", "-n", net, "--synthetic", n))
    assert len(enc) == int(n) and enc == raw


@pytest.mark.gpu
def test_cryptonets_program_encrypted_scores_like_raw():
    enc = run("cryptonets.py", "--synthetic", "300")
    raw = run("cryptonets.py", "--synthetic", "300", "--raw")
    pick = lambda s: re.findall(r"errs (\d+)/300 accuracy", s)[-1]
    assert pick(enc) == pick(raw)


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
