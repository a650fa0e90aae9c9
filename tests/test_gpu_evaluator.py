"""GPU parity: every evaluator entry point of the C ABI vs the CPU oracle, BIT-EXACT (u64 words).

Inputs are fresh BFV encryptions made by the oracle with a seeded RNG; the same words go to both sides.
Tolerance: none - integer arithmetic, every ciphertext word must match.
"""
import numpy as np
import pytest

from conftest import PARAMS, get_gpu, get_oracle

pytestmark = pytest.mark.gpu


def enc_batch(o, rng, count):
    vals = rng.integers(0, o.t, size=(count, o.n), dtype=np.uint64)
    return vals, np.stack([o.encrypt(o.encode(v)) for v in vals])


def up(g, cts, size=2):
    h = g.ct_alloc(len(cts), size)
    g.ct_upload(h, 0, cts)
    return h


@pytest.mark.parametrize("name", ["tiny", "default4096", "c2", "c3"])
def test_ntt_roundtrip_and_parity(name, rng):
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    cts = np.stack([rng.integers(0, q, size=o.n, dtype=np.uint64) for _ in range(2) for q in o.q] * 3).reshape(3, -1)
    h = up(g, cts)
    g.ct_ntt(h, 0, 3)
    got = g.ct_download(h, 0, 3)
    exp = np.stack([np.concatenate([o.ntt_fwd(j % o.k, c.reshape(2 * o.k, o.n)[j]) for j in range(2 * o.k)]) for c in cts])
    assert np.array_equal(got, exp)
    g.ct_ntt(h, 0, 3, inverse=True)
    assert np.array_equal(g.ct_download(h, 0, 3), cts)
    g.free(h)


@pytest.mark.parametrize("name", ["tiny", "default4096", "c3"])
def test_linear_ops(name, rng):
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    vals, cts = enc_batch(o, rng, 4)
    h = up(g, cts)
    out = g.ct_alloc(4)
    g.add(h, 0, h, 2, out, 0, 2)
    g.sub(h, 0, h, 2, out, 2, 2)
    got = g.ct_download(out, 0, 4)
    for i in range(2):
        assert np.array_equal(got[i], o.add(cts[i], cts[i + 2]))
        assert np.array_equal(got[2 + i], o.sub(cts[i], cts[i + 2]))
    g.negate(h, 1, out, 0, 1)
    assert np.array_equal(g.ct_download(out, 0, 1)[0], o.negate(cts[1]))
    g.add_many(h, [0, 3, 1, 2], out, 3)
    exp = o.add(o.add(o.add(cts[0], cts[3]), cts[1]), cts[2])
    assert np.array_equal(g.ct_download(out, 3, 1)[0], exp)
    # add_plain / sub_plain with dense plaintexts (upper-half coefficients exercised by uniform values)
    pv = rng.integers(0, o.t, size=(2, o.n), dtype=np.uint64)
    plains = np.stack([o.encode(v) for v in pv])
    ph = g.pt_alloc(2)
    g.pt_upload(ph, 0, plains)
    g.add_plain(h, 0, ph, 0, out, 0, 2)
    g.add_plain(h, 2, ph, 0, out, 2, 2, subtract=True)
    got = g.ct_download(out, 0, 4)
    for i in range(2):
        assert np.array_equal(got[i], o.add_plain(cts[i], plains[i]))
        assert np.array_equal(got[2 + i], o.add_plain(cts[2 + i], plains[i], subtract=True))
    for x in (h, out, ph):
        g.free(x)


@pytest.mark.parametrize("name", ["tiny", "default4096", "c2", "c3"])
def test_multiply_plain_dense_and_scalar(name, rng):
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    vals, cts = enc_batch(o, rng, 3)
    h, out = up(g, cts), g.ct_alloc(3)
    plains = rng.integers(0, o.t, size=(3, o.n), dtype=np.uint64)        # dense polynomial plaintexts
    ph = g.pt_alloc(3)
    g.pt_upload(ph, 0, plains)
    g.mul_plain(h, 0, ph, 0, out, 0, 3)
    got = g.ct_download(out, 0, 3)
    for i in range(3):
        assert np.array_equal(got[i], o.multiply_plain(cts[i], plains[i]))
    g.mul_plain(h, 0, ph, 1, out, 0, 3, pt_stride=0)                      # one plaintext broadcast
    got = g.ct_download(out, 0, 3)
    for i in range(3):
        assert np.array_equal(got[i], o.multiply_plain(cts[i], plains[1]))
    # constant (sparse-format) plaintexts incl. negative weights t-|w| and the largest residue
    sc = np.array([3, o.t - 7, o.t - 1], dtype=np.uint64)
    g.mul_scalar(h, 0, sc, out, 0, 3)
    got = g.ct_download(out, 0, 3)
    for i in range(3):
        assert np.array_equal(got[i], o.multiply_plain(cts[i], sc[i:i + 1]))
    # all-zero plaintext: SEAL throws "plain cannot be zero" -> CN_ERR_ZERO
    from cryptonets_amd._native import CnError
    g.pt_upload(ph, 2, np.zeros(o.n, dtype=np.uint64))
    with pytest.raises(CnError) as e:
        g.mul_plain(h, 0, ph, 2, out, 0, 1)
    assert e.value.code == -4
    with pytest.raises(CnError):
        g.mul_scalar(h, 0, np.zeros(1, dtype=np.uint64), out, 0, 1)
    for x in (h, out, ph):
        g.free(x)


@pytest.mark.parametrize("name", ["tiny", "default4096", "c3"])
def test_scalar_gemm(name, rng):
    """HOT LOOP A (DenseMatrixBySparseVectorMultiply) incl. shared patches, padded taps, zero weights, bias."""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    n_in, K = 7, 5
    vals, cts = enc_batch(o, rng, n_in)
    h = up(g, cts)
    # conv-like: 3 "corners" with their own gather rows, 4 maps each, map-major output order like PoolLayer
    corners = np.array([[0, 1, 2, 3, 4], [2, 3, -1, 5, 6], [6, 5, 4, -1, -1]], dtype=np.int32)
    maps = 4
    O = maps * len(corners)
    idx = np.zeros((O, K), dtype=np.int32)
    W = rng.integers(0, o.t, size=(O, K), dtype=np.uint64)
    W[1, 2] = 0
    W[5, :2] = 0
    W[7] = np.array([1, o.t - 1, 2, o.t - 2, 0], dtype=np.uint64)
    for m in range(maps):
        for c in range(len(corners)):
            idx[m * len(corners) + c] = corners[c]
    bias_vals = rng.integers(0, o.t, size=maps, dtype=np.uint64)
    bias_plain = np.stack([o.encode(np.full(o.n, b, dtype=np.uint64)) for b in bias_vals])
    bh = g.pt_alloc(maps)
    g.pt_upload(bh, 0, bias_plain)
    bias_idx = np.repeat(np.arange(maps, dtype=np.int32), len(corners))
    out = g.ct_alloc(O + 1)
    g.scalar_gemm(h, W, out, 1, idx=idx, bias_pt=bh, bias_idx=bias_idx)
    got = g.ct_download(out, 1, O)
    exp = o.scalar_gemm(cts, W, idx)
    exp = o.add_plain_batch(exp, bias_plain[bias_idx])
    assert np.array_equal(got, exp)
    # dense layer shape (identity gather, one group), no bias, more outputs than the register tile
    O2 = 13
    W2 = rng.integers(0, o.t, size=(O2, n_in), dtype=np.uint64)
    out2 = g.ct_alloc(O2)
    g.scalar_gemm(h, W2, out2, 0)
    assert np.array_equal(g.ct_download(out2, 0, O2), o.scalar_gemm(cts, W2))
    # groups of different sizes (tiled convolution: border tiles hold fewer outputs) - 6, 3 and 1 outputs sharing a gather list,
    # interleaved in the output order, for general and for small signed weights
    rows = [corners[0]] * 6 + [corners[1]] * 3 + [corners[2]]
    perm = rng.permutation(len(rows))
    idx3 = np.stack([rows[i] for i in perm]).astype(np.int32)
    for W3 in (rng.integers(0, o.t, size=(10, K), dtype=np.uint64),
               (rng.integers(-min(500, (o.t - 1) // 2), min(500, (o.t - 1) // 2) + 1, size=(10, K)) % o.t).astype(np.uint64)):
        W3[:, 0] = 1                                              # every output keeps a non-zero term
        out3 = g.ct_alloc(10)
        g.scalar_gemm(h, W3, out3, 0, idx=idx3)
        assert np.array_equal(g.ct_download(out3, 0, 10), o.scalar_gemm(cts, W3, idx3))
        g.free(out3)
    # an output with no non-zero term is an error (reference: AddMany of an empty list)
    from cryptonets_amd._native import CnError
    W2[3] = 0
    with pytest.raises(CnError):
        g.scalar_gemm(h, W2, out2, 0)
    for x in (h, out, out2, bh):
        g.free(x)


@pytest.mark.parametrize("name", ["tiny", "default4096", "c2", "c3"])
def test_multiply_relinearize(name, rng):
    """HOT LOOP B: Evaluator.Multiply (BEHZ) and Relinearize, separately and fused, incl. squaring."""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    vals, cts = enc_batch(o, rng, 4)
    h = up(g, cts)
    out3, out2 = g.ct_alloc(2, 3), g.ct_alloc(4)
    g.multiply(h, 0, h, 2, out3, 0, 2)
    got3 = g.ct_download(out3, 0, 2, size=3)
    exp3 = [o.multiply(cts[i], cts[i + 2]) for i in range(2)]
    for i in range(2):
        assert np.array_equal(got3[i], exp3[i])
    g.relinearize(out3, 0, out2, 0, 2)
    got2 = g.ct_download(out2, 0, 2)
    for i in range(2):
        assert np.array_equal(got2[i], o.relinearize(exp3[i]))
    # fused multiply+relinearize, squaring path (SquareActivation: m.ElementWiseMultiply(m))
    g.mul_relin(h, 0, h, 0, out2, 0, 4)
    got = g.ct_download(out2, 0, 4)
    exp = o.mul_relin_batch(cts, cts)
    assert np.array_equal(got, exp)
    # decrypted slots are the products (sanity that the compared words are a valid ciphertext)
    if name != "c2":      # 2 limbs (86-bit q) with a 39-bit t leave no noise budget for a ct x ct product
        dec = o.decode(o.decrypt(got[0]))
        assert np.array_equal(dec, np.array([(int(a) * int(a)) % o.t for a in vals[0]], dtype=np.uint64))
    # broadcast one operand (PointwiseMultiplySparseDimOne)
    g.mul_relin(h, 0, h, 3, out2, 0, 3, b_stride=0)
    got = g.ct_download(out2, 0, 3)
    for i in range(3):
        assert np.array_equal(got[i], o.relinearize(o.multiply(cts[i], cts[3])))
    for x in (h, out3, out2):
        g.free(x)


@pytest.mark.parametrize("name", ["tiny", "default4096", "c3", "n16k7"])
def test_squaring_fused_kernel_is_the_separate_launches(name, rng):
    """SquareActivation path: forward transforms + tensor + inverse transforms of a squaring run as ONE kernel per base (NTT-form operands
    parked in the output's place, q side read from the ciphertext in place); cn_set_option("sq_fused", 0) selects the separate launches.
    Same words from both and from the oracle, also at an input offset / stride, and the inputs are left untouched."""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    cnt = 5
    vals, cts = enc_batch(o, rng, cnt)
    h = up(g, cts)
    out3, out2 = g.ct_alloc(cnt, 3), g.ct_alloc(cnt)
    exp3 = [o.multiply(c, c) for c in cts]
    got = {}
    for fused in (1, 2, 3, 0):                                   # 2: fused with the NTT-form operand parked in LDS (cn_set_option("sq_lds", 1)); 3: the pipelined
        g.set_option("sq_fused", int(fused > 0))                 # resident kernel (k_square_pipe, N <= 8192) forced for this small count
        g.set_option("sq_lds", int(fused == 2))
        g.set_option("sq_pipe", 2 if fused == 3 else 0)
        g.multiply(h, 1, h, 1, out3, 1, cnt - 1)                 # squares of ciphertexts 1.. (offset into the array)
        got[fused] = g.ct_download(out3, 1, cnt - 1, size=3)
        for i in range(cnt - 1):
            assert np.array_equal(got[fused][i], exp3[i + 1]), (name, fused, i)
        g.mul_relin(h, 0, h, 0, out2, 0, cnt)
        assert np.array_equal(g.ct_download(out2, 0, cnt), o.mul_relin_batch(cts, cts)), (name, fused)
    g.set_option("sq_fused", 1)
    g.set_option("sq_lds", 1)
    g.set_option("sq_pipe", 1)
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2]) and np.array_equal(got[0], got[3])
    assert np.array_equal(g.ct_download(h, 0, cnt), cts)         # operands intact (the q side is read in place)
    for x in (h, out3, out2):
        g.free(x)


@pytest.mark.parametrize("name,cnt", [("tiny", 701), ("c3", 530)])
def test_squaring_in_two_pipelined_halves_gives_the_same_words(name, cnt, rng):
    """cn_set_option("sq_halves", 1): Multiply + Relinearize of >= 512 ciphertexts as two halves over the context's two streams (the second half's Multiply beside
    the first half's key switch) - the words of the one-stream evaluation, for the batched call, at an offset, followed at once by a reader on the context's stream,
    and for the deferred per-ciphertext calls (operand tables)"""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    from bench import uniform_ct_words
    cts = uniform_ct_words(rng, o.q, o.n, cnt)
    h, ref, out, out2 = up(g, cts), g.ct_alloc(cnt), g.ct_alloc(cnt), g.ct_alloc(cnt)
    try:
        g.set_option("sq_halves", 0)
        g.mul_relin(h, 0, h, 0, ref, 0, cnt)
        want = g.ct_download(ref, 0, cnt)
        g.add(ref, 0, ref, 0, ref, 0, cnt)
        want_sum = g.ct_download(ref, 0, cnt)
        g.set_option("sq_halves", 1)
        for rep in range(3):                                             # (the second stream is created at the first call)
            g.mul_relin(h, 0, h, 0, out, 0, cnt)
            g.add(out, 0, out, 0, out2, 0, cnt)                          # a reader right behind it on the context's stream: ordered behind BOTH halves
            assert np.array_equal(g.ct_download(out2, 0, cnt), want_sum), rep
            assert np.array_equal(g.ct_download(out, 0, cnt), want), rep
        g.mul_relin(h, 3, h, 3, out, 0, cnt - 3)
        assert np.array_equal(g.ct_download(out, 0, cnt - 3), want[3:])
        for i in (0, cnt // 2 - 1, cnt // 2, cnt // 2 + 8, cnt - 1):
            assert np.array_equal(want[i], o.relinearize(o.multiply(cts[i], cts[i]))), i
        assert np.array_equal(g.ct_download(h, 0, cnt), cts)
        hs = [g.ct_alloc(1) for _ in range(cnt)]
        for i, x in enumerate(hs):
            g.copy(h, i, x, 0, 1)
        g.set_option("sq_halves", 2)                                     # ... and inside the flush of queued per-ciphertext calls
        for mode in (1, 2):
            g.set_option("defer", mode)
            rs = [g.ct_alloc(1) for _ in range(cnt)]
            for x, r in zip(hs, rs):
                g.mul_relin(x, 0, x, 0, r, 0, 1)
            g.set_option("defer", 0)
            got = np.stack([g.ct_download(r, 0, 1)[0] for r in rs])
            assert np.array_equal(got, want), mode
            g.free_many(rs)
        g.free_many(hs)
    finally:
        g.set_option("sq_halves", 1)
        g.set_option("defer", 0)
        for x in (h, ref, out, out2):
            g.free(x)


@pytest.mark.parametrize("name,cnt", [("tiny", 700), ("default4096", 300), ("c3", 230)])
def test_pipelined_squaring_of_a_batch(name, cnt, rng):
    """k_square_pipe as the library picks it by itself (a batch of at least four blocks per resident workgroup): every workgroup squares several
    ciphertexts of one limb in turn with the next operand prefetched - uneven shares (cnt is not a multiple of the workgroups per limb), an offset
    into the array, operands at the edges of the residue range, the deferred per-ciphertext form (operand address table) - all against the
    separate launches, a sample against the oracle."""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    from bench import uniform_ct_words
    cts = uniform_ct_words(rng, o.q, o.n, cnt)
    for j, qj in enumerate(o.q):                                     # extreme residues in the first ciphertexts
        w = cts[:3].reshape(3, 2, o.k, o.n)
        w[0, :, j, :] = qj - 1
        w[1, :, j, :] = 0
        w[2, 0, j, :] = qj // 2
    h, out3, ref3, out2 = up(g, cts), g.ct_alloc(cnt, 3), g.ct_alloc(cnt, 3), g.ct_alloc(cnt)
    try:
        g.set_option("sq_pipe", 0)
        g.multiply(h, 0, h, 0, ref3, 0, cnt)
        want = g.ct_download(ref3, 0, cnt, size=3)
        g.set_option("sq_pipe", 1)
        g.multiply(h, 0, h, 0, out3, 0, cnt)
        assert np.array_equal(g.ct_download(out3, 0, cnt, size=3), want)
        g.multiply(h, 7, h, 7, out3, 0, cnt - 7)                     # offset: another share per workgroup
        assert np.array_equal(g.ct_download(out3, 0, cnt - 7, size=3), want[7:])
        for i in (0, 1, 2, 3, cnt // 2, cnt - 1):
            assert np.array_equal(want[i], o.multiply(cts[i], cts[i])), i
        assert np.array_equal(g.ct_download(h, 0, cnt), cts)
        # deferred per-ciphertext calls: every operand its own array, merged into one launch chain with an operand address table
        g.mul_relin(h, 0, h, 0, out2, 0, cnt)
        want2 = g.ct_download(out2, 0, cnt)
        hs = [g.ct_alloc(1) for _ in range(cnt)]
        for i, x in enumerate(hs):
            g.copy(h, i, x, 0, 1)
        g.set_option("defer", 1)
        rs = [g.ct_alloc(1) for _ in range(cnt)]
        for x, r in zip(hs, rs):
            g.mul_relin(x, 0, x, 0, r, 0, 1)
        g.set_option("defer", 0)
        got2 = np.stack([g.ct_download(r, 0, 1)[0] for r in rs])
        assert np.array_equal(got2, want2)
        g.free_many(hs + rs)
    finally:
        g.set_option("defer", 0)
        g.set_option("sq_pipe", 1)
        for x in (h, out3, ref3, out2):
            g.free(x)


@pytest.mark.parametrize("name", ["tiny", "c4", "n16k7"])
def test_multiply_plain_fused_kernels_are_the_separate_launches(name, rng):
    """Dense MultiplyPlain: k_lift_ntt + k_mul_plain_fused (default) against the six separate launches (cn_set_option("mp_fused", 0)) and
    the oracle - one plaintext per ciphertext, one plaintext shared by all, in place, size-3 ciphertexts, and the broadcast form of
    cn_rowdot_batch (ONE input ciphertext against many rows)."""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    cnt = 3
    vals, cts = enc_batch(o, rng, cnt)
    pts = np.stack([o.encode(rng.integers(0, o.t, size=o.n, dtype=np.uint64)) for _ in range(cnt)])
    ph = g.pt_alloc(cnt)
    g.pt_upload(ph, 0, pts)
    exp = [o.multiply_plain(cts[i], pts[i]) for i in range(cnt)]
    exp_shared = [o.multiply_plain(cts[i], pts[1]) for i in range(cnt)]
    for fused in (1, 0):
        g.set_option("mp_fused", fused)
        h, out = up(g, cts), g.ct_alloc(cnt)
        g.mul_plain(h, 0, ph, 0, out, 0, cnt)
        assert np.array_equal(g.ct_download(out, 0, cnt), exp), (name, fused)
        g.mul_plain(h, 0, ph, 1, out, 0, cnt, pt_stride=0)
        assert np.array_equal(g.ct_download(out, 0, cnt), exp_shared), (name, fused, "shared plaintext")
        g.mul_plain(h, 1, ph, 0, h, 1, 2)                                       # in place, at an offset
        assert np.array_equal(g.ct_download(h, 0, cnt), [cts[0], o.multiply_plain(cts[1], pts[0]), o.multiply_plain(cts[2], pts[1])]), (name, fused)
        g.ct_upload(h, 0, cts)
        # size-3 ciphertext x plaintext (the reference never relinearises before a MultiplyPlain in PointwiseMultiply chains)
        h3 = g.ct_alloc(1, 3)
        g.multiply(h, 0, h, 1, h3, 0, 1)
        g.mul_plain(h3, 0, ph, 2, h3, 0, 1)
        assert np.array_equal(g.ct_download(h3, 0, 1, size=3)[0], o.multiply_plain(o.multiply(cts[0], cts[1]), pts[2])), (name, fused)
        # broadcast: rows of a dense layer against ONE vector (no device copies of the vector in the fused form)
        length = 8
        from cryptonets_amd._native import CnError
        with pytest.raises(CnError):
            g.rowdot_batch(h, 0, ph, 0, cnt, length, h, 0)                      # the API refuses to overwrite the vector
        g.rowdot_batch(h, 0, ph, 0, cnt, length, out, 0)
        got = g.ct_download(out, 0, cnt)
        assert np.array_equal(g.ct_download(h, 0, 1)[0], cts[0])
        for r in range(cnt):
            c, sh = o.multiply_plain(cts[0], pts[r]), 1
            while sh < length:                                                 # SumAllSlots(length): RotateRows(-2^s) + Add
                c = o.add(c, o.rotate_rows(c, -sh))
                sh *= 2
            assert np.array_equal(got[r], c), (name, fused, r)
        for x in (h, out, h3):
            g.free(x)
    g.set_option("mp_fused", 1)
    g.free(ph)
    # four rows or more: ONE launch that transforms the plaintexts and multiplies by the once-transformed ciphertext (k_mul_plain_bcast, round 5) - against the
    # oracle and against the two-launch form (cn_set_option("mp_bcast", 0)), rows with a stride in the plaintext array, and the integer transform path
    rows = 5
    pts5 = np.stack([o.encode(rng.integers(0, o.t, size=o.n, dtype=np.uint64)) for _ in range(rows)])
    exp5 = np.stack([o.multiply_plain(cts[1], pts5[r]) for r in range(rows)])
    for gg in (g, get_gpu(name, galois=True, f64=False)):
        ph5, h, out = gg.pt_alloc(rows), up(gg, cts), gg.ct_alloc(rows)
        gg.pt_upload(ph5, 0, pts5)
        for bc in (1, 0):
            gg.set_option("mp_bcast", bc)
            gg.rowdot_batch(h, 1, ph5, 0, rows, 1, out, 0)                      # length 1: the products themselves
            assert np.array_equal(gg.ct_download(out, 0, rows), exp5), (name, bc)
        gg.set_option("mp_bcast", 1)
        assert np.array_equal(gg.ct_download(h, 1, 1)[0], cts[1])
        for x in (ph5, h, out):
            gg.free(x)


@pytest.mark.parametrize("name", ["tiny", "default4096", "c4"])
def test_rotations(name, rng):
    """HOT LOOP C: Galois automorphism + key switch, direct keys and NAF-decomposed steps, column swap."""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    vals, cts = enc_batch(o, rng, 2)
    h, out = up(g, cts), g.ct_alloc(2)
    half = o.n // 2
    for steps in [1, -1, 2, -4, 3, -3, 169, -169, 7, half - 1, -(half - 1), 1024 - half if half > 1024 else 5]:
        if abs(steps) >= half or steps == 0:
            continue
        g.rotate_rows(h, 0, steps, out, 0, 2)
        got = g.ct_download(out, 0, 2)
        for i in range(2):
            assert np.array_equal(got[i], o.rotate_rows(cts[i], steps)), steps
    d = o.decode(o.decrypt(got[0]))
    g.rotate_columns(h, 0, out, 0, 2)
    got = g.ct_download(out, 0, 2)
    for i in range(2):
        assert np.array_equal(got[i], o.rotate_columns(cts[i]))
    assert np.array_equal(o.decode(o.decrypt(got[1])), np.concatenate([vals[1][half:], vals[1][:half]]))
    # in-place rotation (RotateRowsInplace)
    g.rotate_rows(h, 0, -5, h, 0, 1)
    assert np.array_equal(g.ct_download(h, 0, 1)[0], o.rotate_rows(cts[0], -5))
    for x in (h, out):
        g.free(x)


def test_c5_shapes_multiply_and_rotate(rng):
    """N=16384, k=8, dbc=60 (LoLa-CIFAR parameters): one multiply+relinearize and one rotation."""
    o, g = get_oracle("c5", galois=True), get_gpu("c5", galois=True)
    vals, cts = enc_batch(o, rng, 2)
    h, out = up(g, cts), g.ct_alloc(2)
    g.mul_relin(h, 0, h, 1, out, 0, 1)
    assert np.array_equal(g.ct_download(out, 0, 1)[0], o.relinearize(o.multiply(cts[0], cts[1])))
    g.rotate_rows(h, 0, -3, out, 0, 2)
    got = g.ct_download(out, 0, 2)
    for i in range(2):
        assert np.array_equal(got[i], o.rotate_rows(cts[i], -3))
    for x in (h, out):
        g.free(x)


def test_handle_errors_and_leak_counter():
    from cryptonets_amd._native import CnError
    g = get_gpu("tiny", galois=False)
    base = g.live_handles()
    h = g.ct_alloc(2)
    assert g.live_handles() == base + 1
    with pytest.raises(CnError):
        g.add(h, 0, h, 1, h, 2, 1)            # index out of range
    with pytest.raises(CnError):
        g.rotate_rows(h, 0, 1, h, 0, 1)       # no Galois keys in this context
    g.free(h)
    with pytest.raises(CnError):
        g.free(h)
    assert g.live_handles() == base


@pytest.mark.parametrize("name", ["tiny", "c3", "c4"])
def test_integer_transform_path(name, rng):
    """The same ops with the exact-FP64 transforms switched off (the integer Harvey/Shoup path that moduli >= 2^49 and
    the 61-bit BEHZ primes use) - both paths must produce the oracle's words."""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True, f64=False)
    vals, cts = enc_batch(o, rng, 3)
    h, out = up(g, cts), g.ct_alloc(3)
    g.ct_ntt(h, 0, 3)
    exp = np.stack([np.concatenate([o.ntt_fwd(j % o.k, c.reshape(2 * o.k, o.n)[j]) for j in range(2 * o.k)]) for c in cts])
    assert np.array_equal(g.ct_download(h, 0, 3), exp)
    g.ct_ntt(h, 0, 3, inverse=True)
    assert np.array_equal(g.ct_download(h, 0, 3), cts)
    g.mul_relin(h, 0, h, 1, out, 0, 2)
    got = g.ct_download(out, 0, 2)
    for i in range(2):
        assert np.array_equal(got[i], o.relinearize(o.multiply(cts[i], cts[i + 1])))
    g.rotate_rows(h, 0, -7, out, 0, 3)
    got = g.ct_download(out, 0, 3)
    for i in range(3):
        assert np.array_equal(got[i], o.rotate_rows(cts[i], -7))
    plains = rng.integers(0, o.t, size=(1, o.n), dtype=np.uint64)
    ph = g.pt_alloc(1)
    g.pt_upload(ph, 0, plains)
    g.mul_plain(h, 0, ph, 0, out, 0, 3, pt_stride=0)
    got = g.ct_download(out, 0, 3)
    for i in range(3):
        assert np.array_equal(got[i], o.multiply_plain(cts[i], plains[0]))
    for x in (h, out, ph):
        g.free(x)


def test_fp64_path_extreme_values():
    """Worst-case magnitudes for the exact-FP64 butterflies: all-(q-1) and alternating 0/(q-1) coefficient vectors at the
    49-bit primes of the N=16384 set and the 44-bit primes of the N=8192 set."""
    for name in ("c5", "c3"):
        o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
        rows = []
        for pattern in range(3):
            limbs = []
            for _ in range(2):
                for q in o.q:
                    if pattern == 0:
                        limbs.append(np.full(o.n, q - 1, dtype=np.uint64))
                    elif pattern == 1:
                        a = np.zeros(o.n, dtype=np.uint64); a[::2] = q - 1; limbs.append(a)
                    else:
                        a = np.full(o.n, q - 1, dtype=np.uint64); a[: o.n // 2] = 1; limbs.append(a)
            rows.append(np.concatenate(limbs))
        cts = np.stack(rows)
        h = up(g, cts)
        g.ct_ntt(h, 0, 3)
        exp = np.stack([np.concatenate([o.ntt_fwd(j % o.k, c.reshape(2 * o.k, o.n)[j]) for j in range(2 * o.k)]) for c in cts])
        assert np.array_equal(g.ct_download(h, 0, 3), exp)
        g.ct_ntt(h, 0, 3, inverse=True)
        assert np.array_equal(g.ct_download(h, 0, 3), cts)
        out = g.ct_alloc(3)
        g.rotate_rows(h, 0, 1, out, 0, 3)                 # key switch with extreme digits
        got = g.ct_download(out, 0, 3)
        exp_rot = [o.rotate_rows(cts[i], 1) for i in range(3)]
        for i in range(3):
            assert np.array_equal(got[i], exp_rot[i])
        # the same on the BATCH kernels (the fused key switch; at N = 16384 k_keyswitch_pair14 with its lazy accumulators over all 8 digits and the
        # single-recentring forward transform), and a rotate-and-add chain on the extreme words
        try:
            g.set_option("ks_wide", 0)
            g.rotate_rows(h, 0, 1, out, 0, 3)
            assert np.array_equal(g.ct_download(out, 0, 3), np.stack(exp_rot)), name
            g.copy(h, 0, out, 0, 3)
            g.sum_slots(out, 0, 3, 4)
            exp_sum = [o.add(x, o.rotate_rows(x, -2)) for x in [o.add(c, o.rotate_rows(c, -1)) for c in cts]]
            assert np.array_equal(g.ct_download(out, 0, 3), np.stack(exp_sum)), name
        finally:
            g.set_option("ks_wide", -1)
        g.free(h); g.free(out)


@pytest.mark.parametrize("name", ["tiny", "c3", "c5"])
def test_scalar_gemm_small_signed_weights(name, rng):
    """PoolLayer-style weights round(w*scale): small signed integers stored as residues (negative = t - |w|).  These take the
    exact-FP64 limb-split kernel (2x22-bit limbs for <=44-bit q, 3x17-bit for the 49-bit primes); incl. the largest
    admissible magnitude 2^20-1 and K beyond one exact-accumulation window (1024 terms) to exercise the fold."""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    n_in = 6
    vals, cts = enc_batch(o, rng, n_in)
    h = up(g, cts)
    for K, O in ((n_in, 23), (1100 if name == "tiny" else 40, 3)):
        idx = rng.integers(0, n_in, size=(O, K), dtype=np.int32)
        idx[0, 0] = -1
        wmax = min(2 ** 20 - 1, (o.t - 1) // 2)                                    # centred residues: |w| < t/2
        Ws = rng.integers(-min(2 ** 11, wmax), min(2 ** 11, wmax), size=(O, K))
        Ws[0, 1], Ws[1, 0], Ws[2, 2] = wmax, -wmax, 0
        if K > n_in:
            Ws[:, :] = rng.integers(-wmax, wmax + 1, size=(O, K))                   # worst-case magnitudes through the fold
        W = np.where(Ws < 0, o.t + Ws, Ws).astype(np.uint64)
        out = g.ct_alloc(O)
        g.scalar_gemm(h, W, out, 0, idx=idx)
        assert np.array_equal(g.ct_download(out, 0, O), o.scalar_gemm(cts, W, idx)), (name, K)
        g.free(out)
    g.free(h)


@pytest.mark.parametrize("name", ["tiny", "c3"])
def test_scalar_gemm_term_counts_of_the_small_weight_kernel(name, rng):
    """k_scalar_gemm_f64 walks a gather list in sets of four terms, two sets per loop turn and an odd set behind the loop; padded taps and the terms past K
    multiply a valid word by a ZERO the weight table carries (whatever weight the caller passed for a padded tap).  Term counts around the set and pair
    boundaries, 1 / 4 / 5 / 9 outputs per list (register tiles of 1, 5, 5 and 10), padded taps with non-zero weights, against the oracle."""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    n_in = 6
    vals, cts = enc_batch(o, rng, n_in)
    cts[5] = np.concatenate([np.full(o.n, qj - 1, dtype=np.uint64) for _ in range(2) for qj in o.q])
    h = up(g, cts)
    for M in (1, 4, 5, 9):
        for K in (1, 3, 4, 5, 8, 9, 12, 13, 25, 26):
            lists = 2
            O = M * lists
            idx = np.empty((O, K), dtype=np.int32)
            for c in range(lists):
                idx[c::lists] = rng.integers(0, n_in, size=K, dtype=np.int32)
            if K > 1:
                idx[0::lists, K // 2] = -1                             # a padded tap in the first list ...
            Ws = rng.integers(-1000, 1001, size=(O, K))
            Ws[:, 0] = np.where(Ws[:, 0] == 0, 1, Ws[:, 0])           # ... whose (non-zero) weight must not count; every row keeps a real term
            W = np.where(Ws < 0, o.t + Ws, Ws).astype(np.uint64)
            exp = o.scalar_gemm(cts, W, idx)
            out = g.ct_alloc(O + 1)
            g.scalar_gemm(h, W, out, 1, idx=idx)
            assert np.array_equal(g.ct_download(out, 1, O), exp), (name, M, K)
            g.free(out)
    # the one-limb form (words not split: one FMA per term) is chosen when (sum |w| + 1) q_max <= 2^53 for every row: rows AT the bound over inputs whose
    # words are all q_j - 1 (every partial sum as large as it gets), all weights of one sign; and one weight more, where the two-limb form must take over
    room = (1 << 53) // max(int(x) for x in o.q)
    half = (o.t - 1) // 2
    K = 25
    per = min((room - 1) // K, half)
    if per >= 1:
        for extra in (0, 1):
            Ws = np.full((5, K), per, dtype=np.int64)
            Ws[:, 0] += min((room - 1) - per * K, half - per) + extra
            Ws[1] = -Ws[1]
            Ws[3, 1::2] = -Ws[3, 1::2]
            W = np.where(Ws < 0, o.t + Ws, Ws).astype(np.uint64)
            idx = np.full((5, K), 5, dtype=np.int32)                  # the all-(q_j - 1) ciphertext in every tap
            idx[:, 3] = 2
            exp = o.scalar_gemm(cts, W, idx)
            out = g.ct_alloc(6)
            g.scalar_gemm(h, W, out, 1, idx=idx)
            assert np.array_equal(g.ct_download(out, 1, 5), exp), (name, "row sum at the one-limb bound", extra)
            g.free(out)
    g.free(h)


@pytest.mark.parametrize("name", ["tiny", "c3"])
def test_scalar_gemm_pairs_overlapping_gather_lists(name, rng):
    """cn_gemm_plan_create / cn_scalar_gemm merge gather lists that share at least half of their inputs in pairs (the union list, weight 0 for the other list's
    entries - a zero weight is no term): sliding windows of 6 taps at stride 2 (4 shared), 1 / 3 / 5 outputs per window, an odd number of windows, padded taps at
    the border, a 2-D 3 x 3 / stride 1 case, and lists that must NOT be merged (more than five outputs; an input twice in one list) - same words as the oracle and
    as the caller's own lists (cn_set_option("gemm_pair", 0))."""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    n_in = 24
    vals, cts = enc_batch(o, rng, n_in)
    cts[23] = np.concatenate([np.full(o.n, qj - 1, dtype=np.uint64) for _ in range(2) for qj in o.q])
    h = up(g, cts)
    cases = []
    for maps in (1, 3, 5, 6):                                          # 6 outputs per list: left alone
        wins = [[(2 * c + t) if 2 * c + t < n_in else -1 for t in range(6)] for c in range(11)]          # the last windows hang over the border
        cases.append((np.array([w for w in wins for _ in range(maps)], dtype=np.int32), "1-D windows x %d maps" % maps))
    grid = np.arange(20).reshape(4, 5)
    wins = [[int(grid[r + a, c + b]) for a in range(3) for b in range(3)] for r in range(2) for c in range(3)]
    cases.append((np.array([w for w in wins for _ in range(2)], dtype=np.int32), "3 x 3 windows at stride 1"))
    dup = np.array([[0, 1, 2, 1], [1, 2, 3, 4], [1, 2, 3, 4]], dtype=np.int32)
    cases.append((dup, "an input twice in one list"))
    try:
        for idx, what in cases:
            O, K = idx.shape
            Ws = rng.integers(-300, 301, size=(O, K))
            Ws[:, 0] = np.where(Ws[:, 0] == 0, 7, Ws[:, 0])
            Ws[idx < 0] = 99                                           # weights of padded taps do not count
            W = np.where(Ws < 0, o.t + Ws, Ws).astype(np.uint64)
            exp = o.scalar_gemm(cts, W, idx)
            for pair in (1, 0):
                g.set_option("gemm_pair", pair)
                out = g.ct_alloc(O)
                g.scalar_gemm(h, W, out, 0, idx=idx)
                assert np.array_equal(g.ct_download(out, 0, O), exp), (name, what, pair)
                plan = g.gemm_plan(W, idx=idx)
                out2 = g.ct_alloc(O)
                g.gemm_apply(plan, h, out2, 0)
                assert np.array_equal(g.ct_download(out2, 0, O), exp), (name, what, pair, "planned")
                g.free(plan); g.free(out); g.free(out2)
    finally:
        g.set_option("gemm_pair", 1)
    g.free(h)


@pytest.mark.parametrize("name", ["tiny", "c2", "c4"])
def test_scalar_gemm_matrix_core_kernel(name, rng):
    """Scalar GEMMs with >= 16 outputs per gather list run on the int8 matrix cores (k_scalar_gemm_mfma: signed base-256 digits of the
    residues x signed digits of the weights, i32 accumulation, exact FP64 fold).  Against the oracle and against the FP64 kernel
    (cn_set_option("gemm_mfma", 0)): 1, 2 and 3 weight digit planes incl. the extreme digits, output counts around the 32-row tiles and
    beyond four tiles, term counts around the 32-term steps, padded taps, zero weights, bias, two gather lists of different size, and
    the extreme residues 0 / q-1 in the inputs."""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    n_in = 9
    vals, cts = enc_batch(o, rng, n_in)
    cts[7] = np.concatenate([np.full(o.n, qj - 1, dtype=np.uint64) for _ in range(2) for qj in o.q])
    cts[8] = 0
    h = up(g, cts)
    bias_vals = rng.integers(0, o.t, size=3, dtype=np.uint64)
    bias_plain = np.stack([o.encode(np.full(o.n, b, dtype=np.uint64)) for b in bias_vals])
    bh = g.pt_alloc(3)
    g.pt_upload(bh, 0, bias_plain)
    half = (o.t - 1) // 2
    # (18, 100) / (20, 130): 4 and 5 K steps = one turn of the kernel's three-set register ring + 1 / 2 steps left over
    for O, K, wmax in ((16, 5, 127), (33, 32, 128), (100, 33, 32639), (130, 70, 32640), (17, 64, 2 ** 20 - 1), (40, 1, 100), (18, 100, 1000), (20, 130, 77)):
        wmax = min(wmax, half)
        idx = rng.integers(0, n_in, size=(O, K), dtype=np.int32)
        idx[:, :] = idx[0]                                            # one gather list ...
        idx[O // 2:, :] = idx[O // 2]                                 # ... per half
        if O >= 34:
            idx[O // 2:] = np.roll(idx[0], 1)
        if K > 2:
            idx[:, 1] = -1                                            # a padded tap in every list
        Ws = rng.integers(-wmax, wmax + 1, size=(O, K))
        Ws[0, 0], Ws[1, 0], Ws[2, 0] = wmax, -wmax, 0
        if K > 3:
            Ws[3, :] = 0
            Ws[3, 3] = 1
        dead = ~np.any((Ws != 0) & (idx >= 0), axis=1)               # a row without any term is an error on both sides (AddMany of nothing)
        Ws[dead, 0] = 1
        W = np.where(Ws < 0, o.t + Ws, Ws).astype(np.uint64)
        bias_idx = (np.arange(O) % 3).astype(np.int32)
        exp = o.add_plain_batch(o.scalar_gemm(cts, W, idx), bias_plain[bias_idx])
        for mfma in (1, 0):
            g.set_option("gemm_mfma", mfma)
            out = g.ct_alloc(O + 2)
            g.scalar_gemm(h, W, out, 2, idx=idx, bias_pt=bh, bias_idx=bias_idx)
            assert np.array_equal(g.ct_download(out, 2, O), exp), (name, O, K, wmax, mfma)
            g.free(out)
        g.set_option("gemm_mfma", 1)
    g.free(h)
    g.free(bh)


def test_concurrent_callers_one_context(rng):
    """The reference calls the evaluator from Defaults.ThreadCount threads (Utils.cs:46-88): concurrent callers on ONE context
    (ctypes releases the GIL) must all get the oracle's words."""
    import threading
    o, g = get_oracle("tiny", galois=True), get_gpu("tiny", galois=True)
    vals, cts = enc_batch(o, rng, 8)
    h = up(g, cts)
    outs = [g.ct_alloc(2) for _ in range(8)]
    errors = []

    def worker(i):
        try:
            for _ in range(5):
                g.mul_relin(h, i, h, (i + 1) % 8, outs[i], 0, 1)
                g.rotate_rows(h, i, -(i + 1), outs[i], 1, 1)
        except Exception as e:          # noqa
            errors.append(e)
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors
    for i in range(8):
        got = g.ct_download(outs[i], 0, 2)
        assert np.array_equal(got[0], o.relinearize(o.multiply(cts[i], cts[(i + 1) % 8])))
        assert np.array_equal(got[1], o.rotate_rows(cts[i], -(i + 1)))
    for x in outs + [h]:
        g.free(x)


@pytest.mark.parametrize("name,f64", [("tiny", True), ("c3", True), ("c3", False), ("c4", True), ("c5", True), ("default4096", False)])
def test_key_switch_variants_agree(name, f64, rng):
    """The fused one-launch key switch (batches), the two-launch variant that spreads the digit transforms of a few ciphertexts
    over the chip (single-image latency; automatic below 160 (ct, limb) workgroups) and, at N = 16384, the two-halves kernel
    must all give the oracle's words - for relinearisation and for rotations, on the FP64 and on the integer path."""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True, f64=f64)
    vals, cts = enc_batch(o, rng, 3)
    h, out = up(g, cts), g.ct_alloc(3)
    exp_mul = [o.relinearize(o.multiply(cts[i], cts[(i + 1) % 3])) for i in range(3)]
    exp_rot = [o.rotate_rows(c, -3) for c in cts]
    exp_col = [o.rotate_columns(c) for c in cts]
    try:
        # (ks_wide, ks_split14, ks_pair14): fused, two-launch, and - N = 16384 only - the one-launch both-halves kernel (round 5), the
        # two-workgroups-per-limb kernel + combining pass of rounds 1-4, the fused 1024-thread kernel
        for wide, split, pair in ((0, 1, 1), (1, 1, 1), (2, 1, 1)) + (((0, 1, 0), (0, 0, 1)) if o.n == 16384 else ()):
            g.set_option("ks_wide", wide)
            g.set_option("ks_split14", split)
            g.set_option("ks_pair14", pair)
            for i in range(3):
                g.mul_relin(h, i, h, (i + 1) % 3, out, i, 1)
            assert np.array_equal(g.ct_download(out, 0, 3), np.stack(exp_mul)), wide
            g.rotate_rows(h, 0, -3, out, 0, 3)
            assert np.array_equal(g.ct_download(out, 0, 3), np.stack(exp_rot)), wide
            g.rotate_columns(h, 0, out, 0, 3)
            assert np.array_equal(g.ct_download(out, 0, 3), np.stack(exp_col)), wide
    finally:
        g.set_option("ks_wide", -1)
        g.set_option("ks_split14", 1)
        g.set_option("ks_pair14", 1)
    for x in (h, out):
        g.free(x)


@pytest.mark.parametrize("name", ["c5", "n16k7"])
def test_sum_slots_chain_at_n16384(name, rng):
    """SumAllSlots on the batch path at N = 16384 (cn_set_option("ks_wide", 0) forces it for 3 ciphertexts): every link of the rotate-and-add chain
    is ONE launch of k_keyswitch_pair14 that leaves sigma_next(c1) for the next link - against the oracle's rotation-by-rotation sequence, against
    the un-chained links (k_galois_limbs in front of every link) and against the kernels of rounds 1-4; lengths with and without the column swap,
    odd and even numbers of links; a single rotate-and-add with distinct operand / accumulator / result arrays; the XCD-aware block order."""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    n, half = o.n, o.n // 2
    vals, cts = enc_batch(o, rng, 9)

    def reference(c, length):
        ln = length if length else n
        if ln >= half:
            c = o.add(c, o.rotate_columns(c))
            ln = half
        s = 1
        while s < ln:
            c = o.add(c, o.rotate_rows(c, -s))
            s *= 2
        return c
    h = g.ct_alloc(9)
    try:
        g.set_option("ks_wide", 0)
        for length in (0, 8, 4):
            exp2 = [reference(c, length) for c in cts[:2]]
            for pair, chain, xcd in ((1, 1, 0), (1, 0, 0), (0, 1, 0)) + (((1, 1, 1),) if length == 8 else ()):
                g.set_option("ks_pair14", pair); g.set_option("ks_chain", chain); g.set_option("ks_xcd", xcd)
                g.ct_upload(h, 0, cts)
                g.sum_slots(h, 0, 9 if xcd else 2, length)
                exp = np.stack(exp2 + [reference(cts[2], length) if xcd else cts[2]])      # (the third one is outside the range unless all 9 are summed)
                assert np.array_equal(g.ct_download(h, 0, 3), exp), (length, pair, chain, xcd)
                if xcd:                                                        # 8 ciphertexts in the XCD-aware order + 1 in the plain one
                    assert np.array_equal(g.ct_download(h, 8, 1)[0], reference(cts[8], length))
        # the row-dot batch: ONE ciphertext x 5 plaintext rows, then the chain - the product kernel hands the chain its first permuted c1 (no permutation pass at all);
        # against the oracle, with and without the hand-over (mp_bcast 0: two-launch product + k_galois_limbs)
        rows = 5
        pts = np.stack([o.encode(rng.integers(0, o.t, size=n, dtype=np.uint64)) for _ in range(rows)])
        ph, ro = g.pt_alloc(rows), g.ct_alloc(rows)
        g.pt_upload(ph, 0, pts)
        g.set_option("ks_pair14", 1); g.set_option("ks_chain", 1); g.set_option("ks_xcd", 1)
        g.ct_upload(h, 0, cts)
        for length in (0, 8):
            exp_rows = np.stack([reference(o.multiply_plain(cts[4], pts[r]), length) for r in range(rows)])
            for bc in (1, 0):
                g.set_option("mp_bcast", bc)
                l0 = g.stats()["kernel_launches"]
                g.rowdot_batch(h, 4, ph, 0, rows, length, ro, 0)
                launches = g.stats()["kernel_launches"] - l0
                assert np.array_equal(g.ct_download(ro, 0, rows), exp_rows), (length, bc)
                links = 14 if length == 0 else 3
                assert launches == (links + 2 if bc else links + 3), (length, bc, launches)      # transform of the ciphertext + product (or lift + product + permutation) + one launch per link
        g.set_option("mp_bcast", 1)
        assert np.array_equal(g.ct_download(h, 4, 1)[0], cts[4])
        g.free(ph); g.free(ro)
        g.set_option("ks_pair14", 1); g.set_option("ks_chain", 1); g.set_option("ks_xcd", 0)
        g.ct_upload(h, 0, cts)
        from cryptonets_amd._native import CnError
        g.rotate_rows(h, 0, -2, h, 1, 3)                                      # result range = operand range shifted by one ciphertext: the permutation pass reads the
        assert np.array_equal(g.ct_download(h, 1, 3), np.stack([o.rotate_rows(c, -2) for c in cts[:3]]))      # whole operand before anything is written
        assert np.array_equal(g.ct_download(h, 0, 1)[0], cts[0]) and np.array_equal(g.ct_download(h, 4, 5), cts[4:])
        g.ct_upload(h, 0, cts)
        g.rotate_rows(h, 1, 5, h, 0, 3)                                       # ... a multi-hop step count (NAF 4 + 1), shifted the other way: through a staging array
        assert np.array_equal(g.ct_download(h, 0, 3), np.stack([o.rotate_rows(c, 5) for c in cts[1:4]]))
        g.ct_upload(h, 0, cts)
        g.rotate_columns(h, 2, h, 3, 2)
        assert np.array_equal(g.ct_download(h, 3, 2), np.stack([o.rotate_columns(c) for c in cts[2:4]]))
        g.ct_upload(h, 0, cts)
        with pytest.raises(CnError):
            g.rotate_rows_add(h, 4, 1, h, 5, h, 6, 2)                         # accumulator range overlaps the result range with a shift
        assert np.array_equal(g.ct_download(h, 0, 9), cts)
        g.ct_upload(h, 0, cts)
        g.rotate_rows_add(h, 0, -2, h, 1, h, 2, 1)                            # three different arrays
        assert np.array_equal(g.ct_download(h, 2, 1)[0], o.add(cts[1], o.rotate_rows(cts[0], -2)))
        g.rotate_columns_add(h, 0, h, 1, h, 1, 1)                             # accumulator in place
        assert np.array_equal(g.ct_download(h, 1, 1)[0], o.add(cts[1], o.rotate_columns(cts[0])))
        assert np.array_equal(g.ct_download(h, 0, 1)[0], cts[0])
    finally:
        for name_, v in (("ks_wide", -1), ("ks_pair14", 1), ("ks_chain", 1), ("ks_xcd", 1)):
            g.set_option(name_, v)
        g.free(h)


def test_multi_digit_key_switch_at_n16384(rng):
    """The N = 16384 batch kernels with SEVERAL digits per source limb (the reference's N = 16384 networks use one 60-bit digit): three of the CIFAR
    primes with dbc 20 / gdbc 25 (3 and 2 digits per limb) - relinearisation, a rotation and a short SumAllSlots chain on the one-launch kernel
    and on the kernels of rounds 1-4 against the oracle."""
    from cryptonets_amd._native import Context
    from oracle.cno import Oracle
    p = PARAMS["c5"]
    o = Oracle(p["n"], p["t"], q=p["q"][:3], dbc=20, gdbc=25)
    o.keygen(31, galois=True)
    g = Context(p["n"], p["t"], q=p["q"][:3], dbc=20, gdbc=25, device=0)
    g.set_relin_key(o.relin_key())
    for i, e in enumerate(o.galois_elts()):
        g.set_galois_key(e, o.galois_key(i))
    vals, cts = enc_batch(o, rng, 3)
    h, out = up(g, cts), g.ct_alloc(3)
    exp_mul = np.stack([o.relinearize(o.multiply(cts[i], cts[(i + 1) % 3])) for i in range(3)])
    exp_rot = np.stack([o.rotate_rows(c, 5) for c in cts])
    exp_sum = np.stack([o.add(x, o.rotate_rows(x, -2)) for x in [o.add(c, o.rotate_rows(c, -1)) for c in cts]])
    g.set_option("ks_wide", 0)
    for pair in (1, 0):
        g.set_option("ks_pair14", pair)
        for i in range(3):
            g.mul_relin(h, i, h, (i + 1) % 3, out, i, 1)
        assert np.array_equal(g.ct_download(out, 0, 3), exp_mul), pair
        g.rotate_rows(h, 0, 5, out, 0, 3)
        assert np.array_equal(g.ct_download(out, 0, 3), exp_rot), pair
        g.copy(h, 0, out, 0, 3)
        g.sum_slots(out, 0, 3, 4)
        assert np.array_equal(g.ct_download(out, 0, 3), exp_sum), pair
    g.close()


@pytest.mark.parametrize("name,f64,legacy", [("tiny", True, False), ("tiny", True, True), ("c3", True, False), ("c3", False, False), ("c4", True, False),
                                             ("c5", True, False), ("default4096", False, False)])
def test_key_switch_xi_convention_variants_agree(name, f64, legacy, rng):
    """cn_set_option("ks_xi", 1): digits of [c_l (q/q_l)^-1]_{q_l} with keys that carry (q/q_l) 2^(dbc d) s' in every limb - the other
    self-consistent convention of a digit key switch (include/cnhip.h).  Every kernel variant (fused, per-digit, per-source-limb, the
    N = 16384 halves and the fused 1024-thread kernel, the radix-2 LDS kernel; FP64 and integer arithmetic) against the oracle built with
    that convention; keys uploaded in NTT form and - second pass - in coefficient form (cn_load_key form 1)."""
    from cryptonets_amd._native import Context
    from oracle.cno import Oracle
    p = PARAMS[name]
    o = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], ks_xi=True)
    o.keygen(23, galois=True)
    g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
    if not f64:
        g.set_option("f64", 0)
    if legacy:
        g.set_option("legacy_ntt", 1)
    g.set_option("ks_xi", 1)
    assert g.get_option("ks_xi") == 1
    elts = o.galois_elts()
    vals, cts = enc_batch(o, rng, 3)
    h, out = up(g, cts), g.ct_alloc(3)
    exp_mul = [o.relinearize(o.multiply(cts[i], cts[(i + 1) % 3])) for i in range(3)]
    exp_rot = [o.rotate_rows(c, -3) for c in cts]
    exp_col = [o.rotate_columns(c) for c in cts]
    for coeff_form in (False, True):
        if coeff_form:
            g.load_key(0, o.key_to_coeff_form(o.relin_key()), coeff_form=True)
            for i, e in enumerate(elts):
                g.load_key(1, o.key_to_coeff_form(o.galois_key(i)), elt=e, coeff_form=True)
        else:
            g.set_relin_key(o.relin_key())
            for i, e in enumerate(elts):
                g.set_galois_key(e, o.galois_key(i))
        for wide, split, pair in ((0, 1, 1), (1, 1, 1), (2, 1, 1)) + (((0, 1, 0), (0, 0, 1)) if o.n == 16384 else ()):
            g.set_option("ks_wide", wide)
            g.set_option("ks_split14", split)
            g.set_option("ks_pair14", pair)
            for i in range(3):
                g.mul_relin(h, i, h, (i + 1) % 3, out, i, 1)
            assert np.array_equal(g.ct_download(out, 0, 3), np.stack(exp_mul)), (wide, split, pair, coeff_form)
            g.rotate_rows(h, 0, -3, out, 0, 3)
            assert np.array_equal(g.ct_download(out, 0, 3), np.stack(exp_rot)), (wide, split, coeff_form)
            g.rotate_columns(h, 0, out, 0, 3)
            assert np.array_equal(g.ct_download(out, 0, 3), np.stack(exp_col)), (wide, split, coeff_form)
            g.rotate_rows_add(h, 0, 1, h, 0, out, 0, 3)                    # the fused "+ accumulator" forms
            assert np.array_equal(g.ct_download(out, 0, 3), np.stack([o.add(c, o.rotate_rows(c, 1)) for c in cts])), (wide, split, coeff_form)
    # the raw convention on these keys is NOT the client's evaluator (what the start-up self-test notices)
    g.set_option("ks_xi", 0)
    g.set_option("ks_wide", -1)
    g.mul_relin(h, 0, h, 1, out, 0, 1)
    assert not np.array_equal(g.ct_download(out, 0, 1)[0], exp_mul[0])
    for x in (h, out):
        g.free(x)
    g.close()


def _extreme_ciphertexts(o):
    """ciphertext WORDS (not valid encryptions: the multiplication is a function of words) at the corners of the BEHZ argument: every
    coefficient 0, q-1 (= -1), floor(q/2) and ceil(q/2) (the largest centred magnitudes: the tensor product then reaches N (q/2)^2, the
    bound the Shenoy-Kumaresan step must survive), alternating +-floor(q/2), and one limb at q_j - 1 with the others 0"""
    Q = 1
    for qj in o.q:
        Q *= qj
    n = o.n
    pats = [[0] * n, [Q - 1] * n, [Q // 2] * n, [Q // 2 + 1] * n, [(Q // 2) if i % 2 else (Q - Q // 2) for i in range(n)]]
    cts = []
    for pa in pats:
        for pb in (pats[2], pa):
            cts.append(np.concatenate([np.array([x % qj for x in poly], dtype=np.uint64) for poly in (pa, pb) for qj in o.q]))
    lone = np.zeros((2, o.k, n), dtype=np.uint64)
    lone[:, 0, :] = o.q[0] - 1
    cts.append(lone.reshape(-1))
    return np.stack(cts)


@pytest.mark.parametrize("name", ["tiny", "c2", "c4", "n16k7", "c5"])
def test_behz_auxiliary_base_on_extreme_operands(name, monkeypatch):
    """the small (k+1 - at N = 16384, k+2 - primes below 2^49) and SEAL's (61-bit) auxiliary base against the oracle on operands at the
    corners of the base-independence argument: zero, -1, +-q/2 everywhere - products at the Shenoy-Kumaresan bound"""
    from cryptonets_amd._native import Context
    o = get_oracle(name, galois=False)
    p = PARAMS[name]
    cts = _extreme_ciphertexts(o)
    m = len(cts)
    exp3 = np.stack([o.multiply(cts[i], cts[(i + 3) % m]) for i in range(m)])
    exp2 = o.mul_relin_batch(cts, cts)
    for seal_aux in ("1", "0"):
        monkeypatch.setenv("CN_SEAL_AUX", seal_aux)
        g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
        assert g.get_option("behz_small_base") == (seal_aux == "0")
        assert g.get_option("aux_primes") == o.k + (2 if seal_aux == "0" and o.n == 16384 else 1)      # N = 16384: one more small prime
        g.set_relin_key(o.relin_key())
        h, out3, out2 = up(g, cts), g.ct_alloc(m, 3), g.ct_alloc(m)
        for i in range(m):
            g.multiply(h, i, h, (i + 3) % m, out3, i, 1)
        assert np.array_equal(g.ct_download(out3, 0, m, size=3), exp3), seal_aux
        g.mul_relin(h, 0, h, 0, out2, 0, m)                          # squarings: the fused kernel on the FP64 path
        assert np.array_equal(g.ct_download(out2, 0, m), exp2), seal_aux
        g.close()


def test_behz_base_falls_back_when_the_bound_does_not_hold(monkeypatch):
    """k+1 primes below 2^49 are only used when log2 t + log2 N + log2 q + 2 < log2(B m_sk) (cn_build_consts): N = 16384 with seven or
    eight of the CIFAR primes (48-49 bits each) is a few bits short and takes k+2 of them - or, when that is forbidden (CN_AUX_EXTRA=0),
    keeps SEAL's 61-bit base; so does any set with a modulus of 49 bits or more"""
    from cryptonets_amd._native import Context
    monkeypatch.delenv("CN_SEAL_AUX", raising=False)
    for extra in ("1", "0"):
        monkeypatch.setenv("CN_AUX_EXTRA", extra)
        for name, small, more in (("c3", 1, 0), ("c4", 1, 0), ("n16k7", 1, 1), ("c5", 1, 1)):
            p = PARAMS[name]
            g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
            k = g.k
            # the inequality itself, from the primes the context chose
            if more and extra == "0":
                assert g.get_option("behz_small_base") == 0 and g.get_option("aux_primes") == k + 1, name
            else:
                assert g.get_option("behz_small_base") == small and g.get_option("aux_primes") == k + 1 + more, name
            g.close()
    monkeypatch.delenv("CN_AUX_EXTRA")
    g = Context(1024, 12289, q=[0xffffee001, 0x3ffffffffc001], dbc=10, gdbc=20, device=0)      # a 50-bit data prime: integer transforms, SEAL's base
    assert g.get_option("behz_small_base") == 0
    g.close()


@pytest.mark.parametrize("name", ["tiny", "c2", "c3", "c4", "n16k7", "c5"])
def test_behz_auxiliary_base_does_not_change_the_words(name, rng, monkeypatch):
    """libcnhip extends to k+1 auxiliary primes just below 2^49 (exact-FP64 transforms) where SEAL - and the oracle - use k+1 primes
    of 61 bits: the product ciphertext is the same integer polynomial floor(t d / q) - beta reduced mod q_j for ANY sufficiently
    large auxiliary base.  CN_SEAL_AUX=1 selects SEAL's base (integer transforms on the Bsk limbs); both must equal the oracle."""
    from cryptonets_amd._native import Context
    o = get_oracle(name, galois=False)
    p = PARAMS[name]
    vals, cts = enc_batch(o, rng, 3)
    exp = [o.relinearize(o.multiply(cts[i], cts[(i + 1) % 3])) for i in range(3)]
    exp3 = o.multiply(cts[0], cts[1])
    for seal_aux in ("1", "0"):
        monkeypatch.setenv("CN_SEAL_AUX", seal_aux)
        g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
        g.set_relin_key(o.relin_key())
        h, out, out3 = up(g, cts), g.ct_alloc(3), g.ct_alloc(1, 3)
        g.multiply(h, 0, h, 1, out3, 0, 1)
        assert np.array_equal(g.ct_download(out3, 0, 1, size=3)[0], exp3), seal_aux
        for i in range(3):
            g.mul_relin(h, i, h, (i + 1) % 3, out, i, 1)
        assert np.array_equal(g.ct_download(out, 0, 3), np.stack(exp)), seal_aux
        g.mul_relin(h, 0, h, 0, out, 0, 3)                          # squares
        assert np.array_equal(g.ct_download(out, 0, 3), o.mul_relin_batch(cts, cts)), seal_aux
        g.close()


@pytest.mark.parametrize("name", ["tiny", "c4", "c5"])
def test_fused_rotate_and_add(name, rng):
    """cn_rotate_rows_add / cn_rotate_columns_add = rotation followed by cn_add, word for word: direct keys, NAF multi-hop
    steps, step 0, in-place accumulators (agg += rot(c), work = work + rot(work)) and both key-switch variants."""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    vals, cts = enc_batch(o, rng, 4)
    h, out = up(g, cts), g.ct_alloc(2)
    half = o.n // 2
    try:
        for wide in (0, 1, 2):
            g.set_option("ks_wide", wide)
            for steps in (-1, 4, -3, 7 if half > 7 else 3, 0):
                g.rotate_rows_add(h, 0, steps, h, 2, out, 0, 2)
                got = g.ct_download(out, 0, 2)
                for i in range(2):
                    rot = cts[i] if steps == 0 else o.rotate_rows(cts[i], steps)
                    assert np.array_equal(got[i], o.add(cts[i + 2], rot)), (wide, steps)
            g.rotate_columns_add(h, 0, h, 2, out, 0, 2)
            got = g.ct_download(out, 0, 2)
            for i in range(2):
                assert np.array_equal(got[i], o.add(cts[i + 2], o.rotate_columns(cts[i])))
            # in place: out = out + rot(out) (SumAllSlots step), then agg += rot(c)
            g.copy(h, 0, out, 0, 2)
            g.rotate_rows_add(out, 0, -2, out, 0, out, 0, 2)
            exp = [o.add(c, o.rotate_rows(c, -2)) for c in cts[:2]]
            assert np.array_equal(g.ct_download(out, 0, 2), np.stack(exp))
            g.rotate_rows_add(h, 2, -5, out, 0, out, 0, 2)
            exp = [o.add(e, o.rotate_rows(c, -5)) for e, c in zip(exp, cts[2:])]
            assert np.array_equal(g.ct_download(out, 0, 2), np.stack(exp))
    finally:
        g.set_option("ks_wide", -1)
    from cryptonets_amd._native import CnError
    with pytest.raises(CnError):
        g.rotate_rows_add(h, 3, 1, h, 0, out, 0, 2)                # source range runs past the array
    for x in (h, out):
        g.free(x)


@pytest.mark.parametrize("name", ["tiny", "c4"])
def test_sum_slots_and_rowdot_batch(name, rng):
    """cn_sum_slots / cn_rowdot_batch (HOT LOOP C in one call) = the reference's per-row MultiplyPlain, RotateColumns + Add,
    RotateRows(-2^s) + Add sequence, word for word; decrypted: slot sums of v * w_r."""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    n, half, R = o.n, o.n // 2, 3
    vals = rng.integers(0, 50, size=n, dtype=np.uint64)
    ct = o.encrypt(o.encode(vals))
    w = rng.integers(1, 20, size=(R, n), dtype=np.uint64)
    pts = np.stack([o.encode(r) for r in w])
    h, ph, out = g.ct_alloc(1), g.pt_alloc(R), g.ct_alloc(R + 1)
    g.ct_upload(h, 0, ct[None, :])
    g.pt_upload(ph, 0, pts)

    def reference(length):
        res = []
        for r in range(R):
            c = o.multiply_plain(ct, pts[r])
            ln = length if length else n
            if ln >= half:
                c = o.add(c, o.rotate_columns(c))
                ln = half
            s = 1
            while s < ln:
                c = o.add(c, o.rotate_rows(c, -s))
                s *= 2
            res.append(c)
        return np.stack(res)
    for length in (0, half, 8, 1):
        g.rowdot_batch(h, 0, ph, 0, R, length, out, 1)
        got = g.ct_download(out, 1, R)
        assert np.array_equal(got, reference(length)), length
    # full sum: every slot of row r holds sum(v * w_r) mod t
    g.rowdot_batch(h, 0, ph, 0, R, 0, out, 1)
    dec = o.decode(o.decrypt(g.ct_download(out, 1, 1)[0]))
    assert int(dec[0]) == int(np.sum(vals.astype(object) * w[0].astype(object)) % o.t) and len(set(int(x) for x in dec)) == 1
    # sum_slots alone, in place, on a batch of 2
    g.ct_upload(out, 0, np.stack([ct, ct]))
    g.sum_slots(out, 0, 2, 4)
    c = ct
    for s in (1, 2):
        c = o.add(c, o.rotate_rows(c, -s))
    assert np.array_equal(g.ct_download(out, 0, 2), np.stack([c, c]))
    from cryptonets_amd._native import CnError
    with pytest.raises(CnError):
        g.rowdot_batch(out, 1, ph, 0, R, 0, out, 0)                # output range covers the input
    for x in (h, ph, out):
        g.free(x)


def test_scalar_gemm_plan_is_the_one_shot_gemm(rng):
    """cn_gemm_plan_create + cn_gemm_plan_apply (weights resident in HBM, launch only) = cn_scalar_gemm, on fresh inputs, at an output
    offset, for small signed and for general weights; the plan is a handle (leak counter, cn_free)."""
    o, g = get_oracle("c3", galois=False), get_gpu("c3", galois=False)
    vals, cts = enc_batch(o, rng, 6)
    h = up(g, cts)
    idx = np.array([[0, 1, 2, -1], [0, 1, 2, -1], [3, 4, 5, 2], [5, 4, -1, -1], [3, 4, 5, 2]], dtype=np.int32)
    bias_vals = rng.integers(0, o.t, size=2, dtype=np.uint64)
    bias_plain = np.stack([o.encode(np.full(o.n, b, dtype=np.uint64)) for b in bias_vals])
    bh = g.pt_alloc(2)
    g.pt_upload(bh, 0, bias_plain)
    bias_idx = np.array([0, 1, 0, 1, 1], dtype=np.int32)
    live = g.live_handles()
    for W in ((rng.integers(-300, 301, size=(5, 4)) % o.t).astype(np.uint64), rng.integers(1, o.t, size=(5, 4), dtype=np.uint64)):
        W[:, 0] = np.maximum(W[:, 0], 1)
        plan = g.gemm_plan(W, idx=idx, bias_pt=bh, bias_idx=bias_idx)
        out = g.ct_alloc(7)
        exp = o.add_plain_batch(o.scalar_gemm(cts, W, idx), bias_plain[bias_idx])
        g.gemm_apply(plan, h, out, 2)
        assert np.array_equal(g.ct_download(out, 2, 5), exp)
        # the same plan on new inputs
        vals2, cts2 = enc_batch(o, rng, 6)
        g.ct_upload(h, 0, cts2)
        g.gemm_apply(plan, h, out, 0)
        assert np.array_equal(g.ct_download(out, 0, 5), o.add_plain_batch(o.scalar_gemm(cts2, W, idx), bias_plain[bias_idx]))
        g.ct_upload(h, 0, cts)
        from cryptonets_amd._native import CnError
        small = g.ct_alloc(3)
        with pytest.raises(CnError):
            g.gemm_apply(plan, small, out, 0)                      # the plan gathers input 5
        with pytest.raises(CnError):
            g.gemm_apply(plan, h, out, 3)                          # 5 outputs do not fit behind offset 3
        for x in (plan, out, small):
            g.free(x)
    assert g.live_handles() == live
    for x in (h, bh):
        g.free(x)


def test_small_scratch_and_no_pool(rng, monkeypatch):
    """CN_SCRATCH_GB caps the per-call scratch arena: a batch that does not fit is processed in chunks (here: 13 MiB per ciphertext at
    C3 against a 40 MiB cap -> chunks of 3); CN_POOL_GB=0 turns the handle pool off.  Same words either way."""
    from cryptonets_amd._native import Context
    monkeypatch.setenv("CN_SCRATCH_GB", "0.04")
    monkeypatch.setenv("CN_POOL_GB", "0")
    o = get_oracle("c3", galois=False)
    p = PARAMS["c3"]
    g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
    g.set_relin_key(o.relin_key())
    vals, cts = enc_batch(o, rng, 7)
    h, out, out3 = up(g, cts), g.ct_alloc(7), g.ct_alloc(7, 3)
    g.mul_relin(h, 0, h, 0, out, 0, 7)
    assert np.array_equal(g.ct_download(out, 0, 7), o.mul_relin_batch(cts, cts))
    g.multiply(h, 0, h, 1, out3, 0, 6)
    got = g.ct_download(out3, 0, 6, size=3)
    for i in range(6):
        assert np.array_equal(got[i], o.multiply(cts[i], cts[i + 1]))
    for x in (h, out, out3):
        g.free(x)
    tmp = [g.ct_alloc(2) for _ in range(4)]                        # alloc / free cycles without the pool
    for x in tmp:
        g.free(x)
    assert g.live_handles() == 0
    g.close()


@pytest.mark.parametrize("name", ["tiny", "c3"])
def test_scalar_gemm_wide_output_tiles(name, rng):
    """Dense-layer shape: MANY outputs share one gather list, which selects the 20-outputs-per-thread FP64 kernel whose weights travel
    through vector registers and are broadcast inside the FMA (DPP row_newbcast).  37 outputs = one full and one ragged tile, K = 13
    and 1100 (not multiples of the 4-term / 8-term pipeline steps; beyond one exact-accumulation window on the 2x22-bit path), padded
    taps, extreme weights, bias, and a second group with its own gather list."""
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    n_in = 9
    vals, cts = enc_batch(o, rng, n_in)
    h = up(g, cts)
    wmax = min(2 ** 20 - 1, (o.t - 1) // 2)
    for K in (13, 1100 if name == "tiny" else 45):
        O1, O2 = 37, 21
        row1, row2 = rng.integers(0, n_in, size=K, dtype=np.int32), rng.integers(0, n_in, size=K, dtype=np.int32)
        row1[3], row2[0], row2[K - 1] = -1, -1, -1
        idx = np.concatenate([np.tile(row1, (O1, 1)), np.tile(row2, (O2, 1))]).astype(np.int32)
        Ws = rng.integers(-wmax, wmax + 1, size=(O1 + O2, K))
        Ws[0, 0], Ws[1, 1], Ws[2, 1:], Ws[2, 0] = wmax, -wmax, 0, 1          # (an all-zero row is an error: AddMany of nothing)
        W = np.where(Ws < 0, o.t + Ws, Ws).astype(np.uint64)
        bias_vals = rng.integers(0, o.t, size=3, dtype=np.uint64)
        bias_plain = np.stack([o.encode(np.full(o.n, b, dtype=np.uint64)) for b in bias_vals])
        bh = g.pt_alloc(3)
        g.pt_upload(bh, 0, bias_plain)
        bias_idx = rng.integers(0, 3, size=O1 + O2).astype(np.int32)
        out = g.ct_alloc(O1 + O2)
        g.scalar_gemm(h, W, out, 0, idx=idx, bias_pt=bh, bias_idx=bias_idx)
        exp = o.add_plain_batch(o.scalar_gemm(cts, W, idx), bias_plain[bias_idx])
        assert np.array_equal(g.ct_download(out, 0, O1 + O2), exp), (name, K)
        g.free(out); g.free(bh)
    g.free(h)


@pytest.mark.parametrize("name", ["tiny", "c4"])
def test_captured_sequence_replays_on_new_inputs(name, rng):
    """cn_graph_begin / cn_graph_end / cn_graph_launch: a recorded chain (one-shot scalar GEMM with its table upload, squaring +
    relinearisation, rotation, AddMany, temporaries from the handle pool) replayed with ONE launch gives the eager words - on the inputs
    it was recorded with and on new ones written into the same handle.  Synchronising calls are refused while recording."""
    from cryptonets_amd._native import CnError
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    vals, cts = enc_batch(o, rng, 4)
    vals2, cts2 = enc_batch(o, rng, 4)
    W = (rng.integers(-50, 51, size=(3, 4)) % o.t).astype(np.uint64)
    W[:, 0] = np.maximum(W[:, 0], 1)
    h, out = up(g, cts), g.ct_alloc(1)

    def sequence():
        t1, t2 = g.ct_alloc(3), g.ct_alloc(3)
        g.scalar_gemm(h, W, t1, 0)
        g.mul_relin(t1, 0, t1, 0, t2, 0, 3)
        g.rotate_rows(t2, 0, -3, t1, 0, 3)
        g.add_many(t1, [0, 1, 2], out, 0)
        g.free(t1); g.free(t2)

    def expected(c):
        e = o.mul_relin_batch(o.scalar_gemm(c, W), o.scalar_gemm(c, W))
        e = [o.rotate_rows(x, -3) for x in e]
        return o.add(o.add(e[0], e[1]), e[2])

    sequence()                                                   # eager (also warms the pool and the arenas)
    assert np.array_equal(g.ct_download(out, 0, 1)[0], expected(cts))
    live = g.live_handles()
    g.graph_begin()
    with pytest.raises(CnError):
        g.sync()                                                 # refused, the recording goes on
    with pytest.raises(CnError):
        g.ct_download(out, 0, 1)
    sequence()
    graph = g.graph_end()
    assert g.live_handles() == live + 1
    for c in (cts2, cts, cts2):
        g.ct_upload(h, 0, c)
        g.ct_upload(out, 0, cts[:1])                             # stale content must be overwritten by the replay
        g.graph_launch(graph)
        assert np.array_equal(g.ct_download(out, 0, 1)[0], expected(c))
    # eager work between launches may use the pool; the graph's temporaries are reserved
    tmp = [g.ct_alloc(3) for _ in range(3)]
    for x in tmp:
        g.ct_upload(x, 0, cts[:3])
    g.graph_launch(graph)
    assert np.array_equal(g.ct_download(out, 0, 1)[0], expected(cts2))
    for x in tmp:
        assert np.array_equal(g.ct_download(x, 0, 3), cts[:3])
        g.free(x)
    g.free(graph)
    with pytest.raises(CnError):
        g.graph_launch(graph)
    for x in (h, out):
        g.free(x)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "c3"])
def test_key_switch_xcd_placement_is_only_a_placement(name, rng):
    """cn_set_option("ks_xcd", 1): the k workgroups of a ciphertext get block ids of one residue class mod 8 (one XCD).  A pure relabelling of
    (ciphertext, limb) -> block id: 19 ciphertexts = two full groups of 8 through the remap + 3 through the plain tail, words equal to the
    oracle's and to the unmapped launch (fused kernel forced: the automatic choice would take the two-launch variant for so few)"""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    vals, cts = enc_batch(o, rng, 19)
    h, out = up(g, cts), g.ct_alloc(19)
    want = o.mul_relin_batch(cts, cts)
    rot = np.stack([o.rotate_rows(c, 1) for c in cts])
    default_order = g.get_option("ks_xcd")
    try:
        g.set_option("ks_wide", 0)
        for xcd in (1, 2, 0):
            g.set_option("ks_xcd", xcd)
            g.mul_relin(h, 0, h, 0, out, 0, 19)
            assert np.array_equal(g.ct_download(out, 0, 19), want), xcd
            g.rotate_rows(h, 0, 1, out, 0, 19)
            assert np.array_equal(g.ct_download(out, 0, 19), rot), xcd
    finally:
        g.set_option("ks_wide", -1)
        g.set_option("ks_xcd", default_order)
    g.free(h), g.free(out)


@pytest.mark.parametrize("name", ["tiny", "c2", "c3"])
def test_scalar_gemm_on_unrelinearized_products(name, rng):
    """Evaluator.MultiplyPlain / Add take ciphertexts of any size: a scalar GEMM over size-3 ciphertexts (products that have not been
    relinearized) through all three kernels - general weights (128-bit integer accumulation), small signed weights with few outputs
    (exact FP64) and with >= 16 outputs per gather list (int8 matrix cores) - with a bias (lands in c0 only), then ONE Relinearize per
    output; words against the oracle running the same sequence; mixed sizes are refused."""
    from cryptonets_amd._native import CnError
    o, g = get_oracle(name, galois=False), get_gpu(name, galois=False)
    n_in = 6
    vals, cts = enc_batch(o, rng, n_in)
    prod = np.stack([o.multiply(cts[i], cts[(i + 1) % n_in]) for i in range(n_in)])      # size 3
    h3 = up(g, prod, size=3)
    bias_vals = rng.integers(0, o.t, size=2, dtype=np.uint64)
    bias_plain = np.stack([o.encode(np.full(o.n, b, dtype=np.uint64)) for b in bias_vals])
    bh = g.pt_alloc(2)
    g.pt_upload(bh, 0, bias_plain)
    half = (o.t - 1) // 2
    for O, K, wmax in ((7, 4, None), (5, 6, min(1000, half)), (40, 6, min(32000, half))):
        idx = rng.integers(0, n_in, size=(O, K), dtype=np.int32)
        idx[:, :] = idx[0]
        idx[:, 1] = -1
        if wmax is None:
            W = rng.integers(1, o.t, size=(O, K), dtype=np.uint64)
        else:
            Ws = rng.integers(-wmax, wmax + 1, size=(O, K))
            Ws[:, 0] = wmax
            W = np.where(Ws < 0, o.t + Ws, Ws).astype(np.uint64)
        bias_idx = (np.arange(O) % 2).astype(np.int32)
        out3, out2 = g.ct_alloc(O, 3), g.ct_alloc(O)
        g.scalar_gemm(h3, W, out3, 0, idx=idx, bias_pt=bh, bias_idx=bias_idx)
        exp3 = o.add_plain_batch(o.scalar_gemm(prod, W, idx), bias_plain[bias_idx])
        assert np.array_equal(g.ct_download(out3, 0, O, size=3), exp3), (name, O)
        g.relinearize(out3, 0, out2, 0, O)
        assert np.array_equal(g.ct_download(out2, 0, O), np.stack([o.relinearize(c) for c in exp3])), (name, O)
        with pytest.raises(CnError):
            g.scalar_gemm(h3, W, out2, 0, idx=idx)                        # size-3 inputs into size-2 outputs
        g.free(out3)
        g.free(out2)
    g.free(h3)
    g.free(bh)


def test_copy_many_gathers_with_one_launch(rng):
    """cn_copy_many: single ciphertexts (and dense plaintexts) of many arrays into consecutive places of one array - immediate (one launch) and
    queued under deferred submission; shape / range / overlap errors"""
    from cryptonets_amd._native import CnError
    o, g = get_oracle("tiny", galois=False), get_gpu("tiny", galois=False)
    vals, cts = enc_batch(o, rng, 6)
    hs = [up(g, cts[2 * i:2 * i + 2]) for i in range(3)]                 # three arrays of two ciphertexts
    dst = g.ct_alloc(5)
    l0 = g.stats()["kernel_launches"]
    g.copy_many([hs[2], hs[0], hs[1], hs[0]], [1, 0, 1, 1], dst, 1)
    assert g.stats()["kernel_launches"] - l0 == 1
    got = g.ct_download(dst, 1, 4)
    assert np.array_equal(got, np.stack([cts[5], cts[0], cts[3], cts[1]]))
    g.set_option("defer", 1)
    try:
        g.copy_many([hs[1], hs[2]], [0, 0], dst, 0)
        g.add(dst, 0, dst, 1, dst, 4, 1)                                 # reads a queued copy
        assert g.get_option("pending_calls") == 3
        got = g.ct_download(dst, 0, 5)
    finally:
        g.set_option("defer", 0)
    assert np.array_equal(got[0], cts[2]) and np.array_equal(got[1], cts[4]) and np.array_equal(got[4], o.add(cts[2], cts[4]))
    pv = rng.integers(0, o.t, size=(3, o.n), dtype=np.uint64)
    ph = [g.pt_alloc(1) for _ in range(3)]
    for h, v in zip(ph, pv):
        g.pt_upload(h, 0, v[None, :])
    pd = g.pt_alloc(3)
    g.copy_many(ph[::-1], [0, 0, 0], pd, 0)
    assert np.array_equal(g.pt_download(pd, 0, 3), pv[::-1])
    with pytest.raises(CnError):
        g.copy_many([hs[0], ph[0]], [0, 0], dst, 0)                      # plaintext into a ciphertext array
    with pytest.raises(CnError):
        g.copy_many([hs[0]], [2], dst, 0)                                # source index out of range
    with pytest.raises(CnError):
        g.copy_many([hs[0], hs[1]], [0, 0], dst, 4)                      # destination range out of range
    with pytest.raises(CnError):
        g.copy_many([dst, hs[0]], [1, 0], dst, 0)                        # a source inside the destination range
    for h in hs + ph + [dst, pd]:
        g.free(h)


@pytest.mark.parametrize("name", ["tiny", "c4", "c5"])
def test_rotation_with_and_without_the_permutation_pass(name, rng):
    """Small batches rotate through the two-launch key switch, whose kernels apply the automorphism while they load c1 / c0
    (cn_set_option("ks_perm_fused", 1), default) instead of a permutation pass in front of them: same words as with the pass and as the
    oracle - per digit (1-6 ciphertexts) and per source limb (7-32), multi-hop steps, the column swap, in place, and rotate-and-add with the
    accumulator in the output's place (the SumAllSlots step)."""
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    assert g.get_option("ks_perm_fused") == 1
    vals, cts = enc_batch(o, rng, 8)
    h, out = up(g, cts), g.ct_alloc(8)
    half = o.n // 2
    try:
        for count in (1, 3, 8):
            for steps in (1, -2, 7, -(half - 1)):
                exp = np.stack([o.rotate_rows(c, steps) for c in cts[:count]])
                for fused in (1, 0):
                    g.set_option("ks_perm_fused", fused)
                    g.rotate_rows(h, 0, steps, out, 0, count)
                    assert np.array_equal(g.ct_download(out, 0, count), exp), (name, count, steps, fused)
        for fused in (1, 0):
            g.set_option("ks_perm_fused", fused)
            g.rotate_columns(h, 0, out, 0, 2)
            assert np.array_equal(g.ct_download(out, 0, 2), np.stack([o.rotate_columns(c) for c in cts[:2]]))
            w = up(g, cts[:2])
            g.rotate_rows(w, 0, -4, w, 0, 2)                                        # in place
            assert np.array_equal(g.ct_download(w, 0, 2), np.stack([o.rotate_rows(c, -4) for c in cts[:2]]))
            g.ct_upload(w, 0, cts[:2])
            g.rotate_rows_add(w, 0, -1, w, 0, w, 0, 2)                              # x += RotateRows(x, -1), everything in one array
            assert np.array_equal(g.ct_download(w, 0, 2), np.stack([o.add(c, o.rotate_rows(c, -1)) for c in cts[:2]]))
            g.free(w)
    finally:
        g.set_option("ks_perm_fused", 1)
    g.free(h)
    g.free(out)


@pytest.mark.parametrize("name", ["tiny", "c4"])
def test_rotate_rows_many(name, rng):
    """cn_rotate_rows_many: n ciphertexts rotated by n different step counts (direct keys, multi-hop NAF steps, 0) as one launch chain per hop
    round - words of n cn_rotate_rows calls; in place; more rotations than one two-launch key switch takes (run in pieces of the largest size that does); queued
    under deferred submission (rotations of one level share the rounds whatever their step counts); overlapping operands are refused."""
    from cryptonets_amd._native import CnError
    o, g = get_oracle(name, galois=True), get_gpu(name, galois=True)
    half = o.n // 2
    steps = [1, -3, 0, 169 % half, -(half - 1), 2, -676 % half - half, 7]
    vals, cts = enc_batch(o, rng, len(steps))
    exp = np.stack([o.rotate_rows(c, s) if s else c for c, s in zip(cts, steps)])
    h, out = up(g, cts), g.ct_alloc(len(steps) + 1)
    l0 = g.stats()["kernel_launches"]
    g.rotate_rows_many(h, list(range(len(steps))), steps, out, [i + 1 for i in range(len(steps))])
    many = g.stats()["kernel_launches"] - l0
    assert np.array_equal(g.ct_download(out, 1, len(steps)), exp)
    l0 = g.stats()["kernel_launches"]
    for i, s in enumerate(steps):
        if s:
            g.rotate_rows(h, i, s, out, i + 1, 1)
    assert many < g.stats()["kernel_launches"] - l0                       # fewer dispatches than one call per rotation
    w = up(g, cts)
    g.rotate_rows_many(w, list(range(len(steps))), steps, w, list(range(len(steps))))          # in place
    assert np.array_equal(g.ct_download(w, 0, len(steps)), exp)
    with pytest.raises(CnError):
        g.rotate_rows_many(w, [0, 1], [1, 1], w, [1, 2])                   # result 0 overwrites operand 1
    with pytest.raises(CnError):
        g.rotate_rows_many(h, [0, 1], [1, 2], out, [3, 3])                 # two results in one place
    g.ct_upload(w, 0, cts)
    g.set_option("defer", 1)
    try:
        for i, s in enumerate(steps):
            g.rotate_rows(w, i, s, out, i, 1)                              # per-call rotations by different steps: one level of the queue
        got = g.ct_download(out, 0, len(steps))
    finally:
        g.set_option("defer", 0)
    assert np.array_equal(got, exp)
    # more rotations than a two-launch key switch takes at once
    big = 40
    idx = [i % len(steps) for i in range(big)]
    hb = g.ct_alloc(big)
    l0 = g.stats()["kernel_launches"]
    g.rotate_rows_many(h, idx, [steps[i] for i in idx], hb, list(range(big)))
    assert np.array_equal(g.ct_download(hb, 0, big), exp[idx])
    # ... run as a few table-driven pieces (round 5), not as 40 x hops single-ciphertext key switches of two launches each
    assert g.stats()["kernel_launches"] - l0 < 40
    for x in (h, out, w, hb):
        g.free(x)
