"""CryptoNets-MNIST (BASELINE config 3, `CryptoNets/CryptoNets.cs:12-110`) end to end with the reference's trained
weights (tests/golden/cryptonets_weights.npz) on synthetic MNIST-like images (the dataset is not in the reference repo).

Bar: the decrypted, CRT-joined logits equal an EXACT integer model of the network (same rounding of inputs and weights
as the wrapper) in every one of the 8192 slots x 10 outputs - integer equality, no tolerance."""
import os

import numpy as np
import pytest

from cryptonets_amd import cryptonets_mnist as cm
from cryptonets_amd.layers import EncryptLayer, InputLayer, PoolLayer, SquareActivation

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cryptonets_weights.npz")


def weights():
    w = np.load(GOLD)
    return w["Weights_0"], w["Weights_1"], w["Biases_2"], w["Weights_3"], w["Biases_3"]


def build_network(Factory, images):
    from cryptonets_amd import networks
    reader = InputLayer(images, Scale=16.0, NormalizationFactor=1.0 / 256.0, Factory=Factory)
    net = networks.CryptoNets(Factory, reader, np.load(GOLD), timing=False)          # CryptoNets.cs:19-75
    pools = [p for p in networks._chain(net) if isinstance(p, PoolLayer)][::-1]
    return net, tuple(pools)


def synthetic_images(count, seed=1):
    """MNIST-like sparsity: a pixel is 0 with probability 0.81, else uniform in 1..255."""
    r = np.random.default_rng(seed)
    return np.where(r.random((count, 784)) < 0.81, 0, r.integers(1, 256, size=(count, 784))).astype(float)


def mulmod(a, b, p):
    """(a*b) mod p for uint64 arrays with a,b < p < 2^40 without overflow."""
    b = np.asarray(b, dtype=np.uint64)
    hi = (a * (b >> np.uint64(20))) % p
    return (hi * np.uint64(1 << 20) + a * (b & np.uint64(0xFFFFF))) % p


def int_model_mod_p(x_int, layers, p):
    """The network over Z_p on the scaled integers, all samples at once."""
    p = np.uint64(p)
    act = np.asarray(x_int, dtype=np.uint64) % p
    for li, L in enumerate(layers):
        W = cm.residues(L["W"], int(p))
        bias = np.array([b % int(p) for b in L["bias"]], dtype=np.uint64)
        O, K = W.shape
        out = np.zeros((act.shape[0], O), dtype=np.uint64)
        # group outputs sharing a gather row (conv) to vectorise over the outputs
        rows = {}
        for o in range(O):
            rows.setdefault(L["idx"][o].tobytes(), []).append(o)
        for key, outs in rows.items():
            idx = np.frombuffer(key, dtype=np.int32)
            acc = np.zeros((act.shape[0], len(outs)), dtype=np.uint64)
            for k in range(K):
                if idx[k] < 0:
                    continue
                acc = (acc + mulmod(act[:, idx[k]][:, None], W[outs, k][None, :], p)) % p
            out[:, outs] = (acc + bias[outs][None, :]) % p
        act = mulmod(out, out, p) if li < 2 else out
    return act


def test_layer_tables_match_reference_geometry():
    L = cm.layer_tables(*weights(), conv_tile=1)
    assert L[0]["idx"].shape == (845, 25) and (L[0]["idx"] < 0).sum() == 129 * 5       # padded taps
    assert len(L[1]["W"]) == 100 and len(L[1]["W"][0]) == 845 and len(L[2]["W"]) == 10
    w0, w1, b2, w3, b3 = weights()
    assert L[0]["W"][0][0] == int(round(w0[0] * 32)) and L[0]["bias"][0] == int(round(w0[25] * 16 * 32))
    assert L[1]["W"][3][7] == int(round(w1[100 * 7 + 3] * 1024))                        # transposed dense weights
    assert L[1]["bias"][5] == int(round(b2[5] * 2 ** 28)) and L[2]["bias"][9] == int(round(b3[9] * 2 ** 61))


def test_tiled_conv_tables_are_the_same_sums():
    """conv_tile=2: 2x2 neighbouring output positions share one gather list of <= 49 pixels; every output still has exactly the
    reference's (input, weight) terms - the extra entries carry weight 0."""
    ref, til = cm.layer_tables(*weights(), conv_tile=1)[0], cm.layer_tables(*weights(), conv_tile=2)[0]
    assert til["idx"].shape == (845, 49) and len({r.tobytes() for r in til["idx"]}) == 49
    assert til["bias"] == ref["bias"] and til["scale"] == ref["scale"]
    for o in range(845):
        want = {(int(g), w) for g, w in zip(ref["idx"][o], ref["W"][o]) if g >= 0 and w != 0}
        got = {(int(g), w) for g, w in zip(til["idx"][o], til["W"][o]) if g >= 0 and w != 0}
        assert got == want
        assert all(w == 0 for g, w in zip(til["idx"][o], til["W"][o]) if g < 0)


def test_layer_classes_and_bench_tables_agree():
    """PoolLayer (mirror of the reference class) and cryptonets_mnist.layer_tables (what bench.py runs) derive the same
    integer weights, biases and gather indices."""
    net, (conv, d3, d5) = build_network(None, np.zeros((1, 784)))
    L = cm.layer_tables(*weights(), conv_tile=1)
    for layer, T in zip((conv, d3, d5), L):
        layer.Prepare()
        corners, maps = len(layer.engine.Corners), layer.engine.maps
        for m in range(maps):
            for c in (0, corners - 1):
                o = m * corners + c
                assert list(layer.gather[c]) == list(T["idx"][o])
                assert layer.weightWindows[m] == T["W"][o]


def test_int_model_small_against_python_ints():
    """the vectorised Z_p model equals plain Python big-int arithmetic on a few samples"""
    L = cm.layer_tables(*weights())
    x = np.rint(synthetic_images(3, seed=5) / 256.0 * 16.0).astype(np.int64)
    p = cm.PLAIN_PRIMES[0]
    got = int_model_mod_p(x, L, p)
    for s in range(3):
        act = [int(v) for v in x[s]]
        for li, T in enumerate(L):
            out = []
            for o in range(len(T["W"])):
                acc = T["bias"][o]
                for k, idx in enumerate(T["idx"][o]):
                    if idx >= 0:
                        acc += T["W"][o][k] * act[idx]
                out.append(acc)
            act = [v * v for v in out] if li < 2 else out
        assert [v % p for v in act] == [int(v) for v in got[s]]


@pytest.mark.gpu
def test_cryptonets_mnist_end_to_end_gpu():
    from oracle_backend import make_factory
    Factory = make_factory("gpu", primes=cm.PLAIN_PRIMES, n=cm.N, galois=False)
    env = Factory.AllocateComputationEnv()
    images = synthetic_images(cm.N, seed=1)
    net, _ = build_network(Factory, images)
    net.PrepareNetwork()
    out = net.GetNext()                                            # encrypt -> 5 evaluated layers on the GPU
    assert out.ColumnCount == 10 and out.RowCount == cm.N
    L = cm.layer_tables(*weights())
    x_int = np.rint(images / 256.0 * 16.0).astype(np.int64)
    # per plaintext prime: every slot of every logit equals the Z_p model
    per_prime = []
    for i, e in enumerate(env.Environments):
        model = int_model_mod_p(x_int, L, e.plainmodulusValue)
        got = np.array([out.GetColumn(c).eVectors[i]._decrypt_ints(e) for c in range(10)], dtype=np.uint64).T
        assert np.array_equal(got, model), "prime %d" % e.plainmodulusValue
        per_prime.append(got)
    # CRT join (EncryptedSealBfvVector.cs:381-411) equals exact big-int arithmetic on a subset of samples
    M = env.bigFactor
    full = [out.GetColumn(c).DecryptFullPrecision(env) for c in range(10)]
    for s in range(0, cm.N, 257):
        act = [int(v) for v in x_int[s]]
        for li, T in enumerate(L):
            o_ = [T["bias"][o] + sum(T["W"][o][k] * act[idx] for k, idx in enumerate(T["idx"][o]) if idx >= 0) for o in range(len(T["W"]))]
            act = [v * v for v in o_] if li < 2 else o_
        for c in range(10):
            exp = act[c] % M
            if exp * 2 > M:
                exp -= M
            assert full[c][s] == exp
    # the same numbers through the double path of Decrypt (what the reference prints): argmax agrees with the integer logits
    dec = out.Decrypt(env)
    assert dec.shape == (cm.N, 10)
    ints = np.array(full, dtype=object).T
    assert all(int(np.argmax(dec[s])) == int(np.argmax([int(v) for v in ints[s]])) for s in range(0, cm.N, 97))
    out.Dispose()


def test_dense_exact_model_equals_the_term_by_term_model():
    """bench.py verifies all 8192 slots of the measured batch with cm.model_mod_p_dense (float64 BLAS on 14-bit limbs, exact);
    it must be the same function as the term-by-term integer model and as the unbounded Python-integer model reduced mod p"""
    layers = cm.layer_tables(*weights())
    imgs = synthetic_images(48, seed=77)
    imgs[0, :] = 255.0                                                  # the largest activations the network can see
    x = np.rint(imgs * cm.NORMALIZATION * cm.INPUT_SCALE).astype(np.int64)
    w = dict(zip(("Weights_0", "Weights_1", "Biases_2", "Weights_3", "Biases_3"), weights()))
    for p in cm.PLAIN_PRIMES:
        fast = cm.model_mod_p_dense(x, layers, p)
        assert np.array_equal(fast, cm.model_mod_p(x, layers, p))
        for s in (0, 1, 47):
            assert [int(v) for v in fast[s]] == [v % p for v in cm.int_logits(w, imgs[s])]


@pytest.mark.gpu
def test_relinearize_late_program_decrypts_to_the_same_logits():
    """CryptoNetsChannel.forward_relinearize_late (opt-in, NOT the reference's call sequence): squarings leave size-3 products, the dense
    layers run on them, Relinearize once per dense OUTPUT (110 key switches per channel instead of 945).  Same SEAL operations (their
    words are pinned per operation in test_gpu_evaluator.py::test_scalar_gemm_on_unrelinearized_products); here the whole batch at
    BASELINE config 3: every slot of every logit equals the integer model for both plaintext primes, and equals what the reference's
    sequence decrypts to, while the ciphertext words differ (digit decomposition is not linear)."""
    from cryptonets_amd._native import Context
    layers = cm.layer_tables(*weights())
    x_int = np.rint(cm.synthetic_images(cm.N, seed=9) * cm.NORMALIZATION * cm.INPUT_SCALE).astype(np.int64)
    for p in cm.PLAIN_PRIMES:
        g = Context(cm.N, p, dbc=10, gdbc=20, device=0)
        g.keygen(0x51CE ^ p, galois=False)
        ch = cm.CryptoNetsChannel(g, layers, cm.constant_plaintext(cm.N))
        ph = g.pt_alloc(784)
        g.encode_batch(np.mod(x_int.T, p).astype(np.uint64), ph, 0)
        g.encrypt(ph, 0, ch.h_in, 0, 784, seed=123)
        g.free(ph)

        def logits():
            dh = g.pt_alloc(10)
            g.decrypt(ch.h5, 0, 10, dh, 0)
            got = g.decode_batch(dh, 0, 10).T
            g.free(dh)
            return got

        ch.forward()
        ref_words, ref_logits = g.ct_download(ch.h5, 0, 10), logits()
        g.stats(reset=True)
        ch.forward_relinearize_late()
        late_words, late_logits = g.ct_download(ch.h5, 0, 10), logits()
        assert g.stats()["Relinarization"] == 110                # the reference's spelling (OperationsCount)
        model = cm.model_mod_p_dense(x_int, layers, p)
        assert np.array_equal(ref_logits, model) and np.array_equal(late_logits, model)
        assert not np.array_equal(ref_words, late_words)
        g.close()
