"""CryptoNets-MNIST (the reference's headline workload) expressed as batched libcnhip calls.

Workload definition follows `CryptoNets/CryptoNets.cs:12-110`: 8192-slot batch, plaintext primes
{549764251649, 549764284417}, N = 8192, CoeffModulus128(8192) (5 limbs), dbc 10:
  PoolLayer conv 28x28, 5x5, stride 2, upper pad 1, 5 maps (scale 32; bias = 26th weight of every map)
  -> SquareActivation -> PoolLayer dense 845->100 (scale 1024, Biases_2) -> SquareActivation
  -> PoolLayer dense 100->10 (scale 32, Biases_3).
The timed window is the one the reference's TimingLayers bracket (`CryptoNets.cs:31,74`): after encryption,
before decryption.  One plaintext prime = one independent CRT channel = one device context
(`EncryptedSealBfvVector.cs:225-236` fans every op out per prime).
"""
import numpy as np

from .convolution import ConvolutionEngine

PLAIN_PRIMES = (549764251649, 549764284417)
N = 8192
INPUT_SCALE = 16.0
NORMALIZATION = 1.0 / 256.0
WEIGHT_SCALE = 32


def _round_scaled(values, scale):
    """v.Multiply(Scale).PointwiseRound() -> BigInteger (EncryptedSealBfvVector.cs:355-357): double product,
    round-half-even, exact integer."""
    return [int(round(float(v) * float(scale))) for v in values]


def _corner_tiles(eng, gather, tile):
    """Blocks of tile x tile neighbouring output positions of the convolution and the union of the input pixels they read.
    All outputs of a block share ONE gather list (taps an output does not use get weight 0), so the scalar GEMM reads each
    input ciphertext of the block once for tile^2 * maps outputs: 5x5 windows at stride 2 overlap 3/5 per axis, a 2x2 block
    reads 49 pixels instead of 4 x 25.  Same sums, same ciphertext words - only fewer passes over HBM."""
    coords = [sorted(set(c[d] for c in eng.Corners)) for d in range(len(eng.Corners[0]))]
    tiles = (tile,) * len(coords) if isinstance(tile, int) else tuple(tile)           # per-axis block shape, e.g. (1, 2)
    key = [tuple(coords[d].index(c[d]) // tiles[d] for d in range(len(c))) for c in eng.Corners]
    names = {k: i for i, k in enumerate(sorted(set(key)))}
    tile_of = [names[k] for k in key]
    unions = [sorted({int(g) for c, t in enumerate(tile_of) if t == i for g in gather[c] if g >= 0}) for i in range(len(names))]
    return tile_of, unions


def layer_tables(weights0, weights1, biases2, weights3, biases3, conv_tile=1):
    """Integer (scaled, signed) weight / bias tables and gather indices of the three PoolLayers.

    weights0: 130 doubles (5 maps x 26, last of each 26 is the bias), weights1: 84500 (845x100, transposed by
    CryptoNets.Transpose, CryptoNets.cs:112-123), weights3: 1000 (10x100).  Returns a list of dicts with
    idx [O,K] int32, W [O,K] python ints, bias [O] python ints, in the reference's output order (map-major).
    conv_tile: outputs of conv_tile x conv_tile neighbouring positions share a gather list (1 = one list per position, the
    reference's layout and the default: measured on MI355X the 2x2 tiling halves the input reads but doubles the FMA work of
    the conv layer, 0.69 ms vs 0.59 ms per prime - the layer is not HBM-bound once L2/MALL catch the window overlap)."""
    layers = []
    # --- conv (no-bias branch of PoolLayer.Apply, PoolLayer.cs:196-227): kernelSize = 25 + 1
    eng = ConvolutionEngine([28, 28], [5, 5], [2, 2], Upperpadding=[1, 1], MapCount=[5, 1])
    ks = 26
    gather = eng.gather_table()                                   # [169, 25]
    win = eng.weight_windows(weights0, ks)                        # [5, 25]
    O = eng.maps * len(eng.Corners)
    if conv_tile == 1:                                         # the reference's own lists: one per output position, taps in Offsets order
        tile_of, unions = list(range(len(eng.Corners))), None
        kmax = gather.shape[1]
    else:
        tile_of, unions = _corner_tiles(eng, gather, conv_tile)
        kmax = max(len(u) for u in unions)
    idx = np.full((O, kmax), -1, dtype=np.int32)
    W, bias = [], []
    s_in = INPUT_SCALE
    for m in range(eng.maps):
        wrow = _round_scaled(win[m], WEIGHT_SCALE)
        b = _round_scaled([weights0[(m + 1) * ks - 1]], s_in * WEIGHT_SCALE)[0]
        for c in range(len(eng.Corners)):
            if unions is None:
                idx[m * len(eng.Corners) + c] = gather[c]
                W.append(wrow)
                bias.append(b)
                continue
            u = unions[tile_of[c]]
            pos = {g: i for i, g in enumerate(u)}
            row = [0] * kmax
            for t, g in enumerate(gather[c]):
                if g >= 0:                                     # padded taps contribute nothing (PoolLayer.cs:68-80: Enc(0))
                    row[pos[int(g)]] = wrow[t]
            idx[m * len(eng.Corners) + c, :len(u)] = u
            W.append(row)
            bias.append(b)
    layers.append(dict(idx=idx, W=W, bias=bias, scale=s_in * WEIGHT_SCALE))
    s = (s_in * WEIGHT_SCALE) ** 2
    # --- dense 845 -> 100
    w1t = np.zeros(len(weights1))
    for i in range(845):
        for j in range(100):
            w1t[i + 845 * j] = weights1[100 * i + j]
    ws = WEIGHT_SCALE * WEIGHT_SCALE
    W = [_round_scaled(w1t[m * 845:(m + 1) * 845], ws) for m in range(100)]
    bias = _round_scaled(biases2, s * ws)
    layers.append(dict(idx=np.tile(np.arange(845, dtype=np.int32), (100, 1)), W=W, bias=bias, scale=s * ws))
    s = (s * ws) ** 2
    # --- dense 100 -> 10
    W = [_round_scaled(weights3[m * 100:(m + 1) * 100], WEIGHT_SCALE) for m in range(10)]
    bias = _round_scaled(biases3, s * WEIGHT_SCALE)
    layers.append(dict(idx=np.tile(np.arange(100, dtype=np.int32), (10, 1)), W=W, bias=bias, scale=s * WEIGHT_SCALE))
    return layers


def reference_weights():
    """The reference's TRAINED CryptoNets-MNIST weights (`CryptoNets/Weights.cs:8-978`), shipped as package data
    (`data/cryptonets_weights.npz`, extracted by tests/golden/make_cryptonets_weights.py): (Weights_0, Weights_1, Biases_2, Weights_3,
    Biases_3) in the order layer_tables takes them."""
    import os
    w = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cryptonets_weights.npz"))
    return tuple(w[k] for k in ("Weights_0", "Weights_1", "Biases_2", "Weights_3", "Biases_3"))


def synthetic_weights(seed=1):
    """Random-init weights of the CryptoNets-MNIST architecture (the trained ones live in the reference repo)."""
    r = np.random.default_rng(seed)
    return (r.normal(0, 0.4, 130), r.normal(0, 0.05, 84500), r.normal(0, 0.05, 100), r.normal(0, 0.3, 1000), r.normal(0, 0.1, 10))


def residues(int_rows, p):
    return np.array([[x % p for x in row] for row in int_rows], dtype=np.uint64)


class CryptoNetsChannel:
    """One plaintext-prime channel of the network on one GPU: device buffers + the five layer launches."""

    def __init__(self, ctx, layers, encode_constant):
        """ctx: cryptonets_amd._native.Context for this prime.  encode_constant(value mod t) -> N plaintext
        coefficients of the dense vector holding `value` in every slot (BatchEncoder of a constant vector is the
        constant polynomial, so this is [value, 0, 0, ...])."""
        self.g, p = ctx, ctx.t
        self.layers = []
        for L in layers:
            W = residues(L["W"], p)
            bias = np.array([b % p for b in L["bias"]], dtype=np.uint64)
            uniq, inv = np.unique(bias, return_inverse=True)
            bh = ctx.pt_alloc(len(uniq))
            ctx.pt_upload(bh, 0, np.stack([encode_constant(int(v)) for v in uniq]))
            # the layer's weights are planned once and stay in HBM (cn_gemm_plan_create); forward() only launches
            plan = ctx.gemm_plan(W, idx=L["idx"], bias_pt=bh, bias_idx=inv.astype(np.int32))
            self.layers.append(dict(idx=L["idx"], W=W, bias_pt=bh, bias_idx=inv.astype(np.int32), plan=plan))
        self.h_in = ctx.ct_alloc(784)
        self.h1, self.h2 = ctx.ct_alloc(845), ctx.ct_alloc(845)
        self.h3, self.h4 = ctx.ct_alloc(100), ctx.ct_alloc(100)
        self.h5 = ctx.ct_alloc(10)

    def forward(self):
        from . import tracing                      # roctx ranges per layer (no-ops unless CN_ROCTX=1)
        g, L = self.g, self.layers
        with tracing.range("PoolLayer conv 5x5 s2 x5 (784 -> 845)", sync=g.sync):
            g.gemm_apply(L[0]["plan"], self.h_in, self.h1, 0)
        with tracing.range("SquareActivation 845", sync=g.sync):
            g.mul_relin(self.h1, 0, self.h1, 0, self.h2, 0, 845)
        with tracing.range("PoolLayer dense 845 -> 100", sync=g.sync):
            g.gemm_apply(L[1]["plan"], self.h2, self.h3, 0)
        with tracing.range("SquareActivation 100", sync=g.sync):
            g.mul_relin(self.h3, 0, self.h3, 0, self.h4, 0, 100)
        with tracing.range("PoolLayer dense 100 -> 10", sync=g.sync):
            g.gemm_apply(L[2]["plan"], self.h4, self.h5, 0)

    # The same five layers in two halves, for a host that STAGGERS the plaintext-prime channels (bench.py --stagger): front() ends where the
    # long FP64-bound kernel of the batch - the key switch of the 845-ciphertext squaring layer - begins.  Same kernels, same words.
    def front(self):
        g, L = self.g, self.layers
        if not hasattr(self, "t3"):
            self.t3 = g.ct_alloc(845, 3)
        g.gemm_apply(L[0]["plan"], self.h_in, self.h1, 0)
        g.multiply(self.h1, 0, self.h1, 0, self.t3, 0, 845)          # Evaluator.Multiply of SquareActivation (AtomicSealBfvVector.cs:839)

    def back(self):
        g, L = self.g, self.layers
        g.relinearize(self.t3, 0, self.h2, 0, 845)                   # Evaluator.Relinearize (AtomicSealBfvVector.cs:840)
        g.gemm_apply(L[1]["plan"], self.h2, self.h3, 0)
        g.mul_relin(self.h3, 0, self.h3, 0, self.h4, 0, 100)
        g.gemm_apply(L[2]["plan"], self.h4, self.h5, 0)

    # experiment (bench.py --stagger 2): the split behind the whole squaring layer - cn_mul_relin pipelines its 845 ciphertexts in parts by itself ("sq_halves")
    def front2(self):
        g, L = self.g, self.layers
        g.gemm_apply(L[0]["plan"], self.h_in, self.h1, 0)
        g.mul_relin(self.h1, 0, self.h1, 0, self.h2, 0, 845)

    def back2(self):
        g, L = self.g, self.layers
        g.gemm_apply(L[1]["plan"], self.h2, self.h3, 0)
        g.mul_relin(self.h3, 0, self.h3, 0, self.h4, 0, 100)
        g.gemm_apply(L[2]["plan"], self.h4, self.h5, 0)


    # NOT the reference's call sequence (opt-in; `bench.py` reports it beside the headline, never as the headline): the squarings leave their
    # products UNRELINEARIZED (size 3) and the dense layer behind them runs on size-3 ciphertexts - Evaluator.MultiplyPlain / Add accept any
    # size - so that Relinearize runs once per OUTPUT of the dense layer (100 + 10 key switches per channel) instead of once per input
    # (845 + 100: PointwiseMultiply relinearizes at once, AtomicSealBfvVector.cs:839-840).  Every step is the same SEAL operation on the same
    # kernels and the words are those the oracle produces for THIS sequence (tests/test_cryptonets_mnist.py); they differ from the
    # reference's sequence (digit decomposition is not linear), the decrypted logits do not.
    def forward_relinearize_late(self):
        g, L = self.g, self.layers
        if not hasattr(self, "t3"):
            self.t3 = g.ct_alloc(845, 3)
        if not hasattr(self, "u3"):
            self.u3, self.v3, self.w3 = g.ct_alloc(100, 3), g.ct_alloc(100, 3), g.ct_alloc(10, 3)
        g.gemm_apply(L[0]["plan"], self.h_in, self.h1, 0)
        g.multiply(self.h1, 0, self.h1, 0, self.t3, 0, 845)
        g.gemm_apply(L[1]["plan"], self.t3, self.u3, 0)                # dense 845 -> 100 on size-3 ciphertexts (bias lands in c0)
        g.relinearize(self.u3, 0, self.h3, 0, 100)
        g.multiply(self.h3, 0, self.h3, 0, self.v3, 0, 100)
        g.gemm_apply(L[2]["plan"], self.v3, self.w3, 0)
        g.relinearize(self.w3, 0, self.h5, 0, 10)


def constant_plaintext(n):
    def enc(v):
        p = np.zeros(n, dtype=np.uint64)
        p[0] = v
        return p
    return enc


# ------------------------------------------------------------------ exact functional model (plain integers mod p)
def mulmod_u64(a, b, p):
    """(a*b) mod p for uint64 arrays with a, b < p < 2^40 (no overflow: 20-bit split of b)"""
    b = np.asarray(b, dtype=np.uint64)
    hi = (a * (b >> np.uint64(20))) % p
    return (hi * np.uint64(1 << 20) + a * (b & np.uint64(0xFFFFF))) % p


def model_mod_p(x_int, layers, p):
    """The network over Z_p on the scaled integer inputs x_int [samples, 784]: what every slot of the decrypted logits of the
    plaintext-prime channel p must equal (conv+bias, square, dense+bias, square, dense+bias)."""
    p = np.uint64(p)
    act = np.asarray(x_int, dtype=np.int64)
    act = np.mod(act, int(p)).astype(np.uint64)
    for li, L in enumerate(layers):
        W = residues(L["W"], int(p))
        bias = np.array([b % int(p) for b in L["bias"]], dtype=np.uint64)
        O, K = W.shape
        out = np.zeros((act.shape[0], O), dtype=np.uint64)
        rows = {}
        for o in range(O):
            rows.setdefault(L["idx"][o].tobytes(), []).append(o)
        for key, outs in rows.items():
            idx = np.frombuffer(key, dtype=np.int32)
            acc = np.zeros((act.shape[0], len(outs)), dtype=np.uint64)
            for k in range(K):
                if idx[k] >= 0:
                    acc = (acc + mulmod_u64(act[:, idx[k]][:, None], W[outs, k][None, :], p)) % p
            out[:, outs] = (acc + bias[outs][None, :]) % p
        act = mulmod_u64(out, out, p) if li < 2 else out
    return act


def model_mod_p_dense(x_int, layers, p):
    """model_mod_p for MANY samples (bench.py checks all 8192 slots of the measured batch): the same exact network over Z_p, every layer
    as one dense integer matrix product evaluated exactly in float64 BLAS - the activations (< p < 2^40) are split into 14-bit limbs,
    limb (< 2^14) x weight (|w| < 2^28) x K (<= 2^11 terms) stays below 2^53, so every partial sum is an exact integer; the limbs are
    recombined mod p in integer arithmetic.  tests/test_cryptonets_mnist.py holds it equal to model_mod_p."""
    P = int(p)
    pu = np.uint64(P)
    act = np.mod(np.asarray(x_int, dtype=np.int64), P).astype(np.uint64)
    for li, L in enumerate(layers):
        idx = np.asarray(L["idx"])
        O, K = idx.shape
        Wi = np.array(L["W"], dtype=np.int64)
        if np.abs(Wi).max() >= 1 << 28 or K > 2048:
            raise ValueError("weights / term count beyond the exact-float64 range of this model")
        dense = np.zeros((O, act.shape[1]), dtype=np.float64)
        rows, cols = np.nonzero(idx >= 0)
        np.add.at(dense, (rows, idx[rows, cols]), Wi[rows, cols].astype(np.float64))
        out = np.zeros((act.shape[0], O), dtype=np.uint64)
        for limb in range(3):
            part = ((act >> np.uint64(14 * limb)) & np.uint64(0x3FFF)).astype(np.float64) @ dense.T          # exact integers, |.| < 2^53
            part = np.mod(part.astype(np.int64), P).astype(np.uint64)
            out = (out + mulmod_u64(part, np.uint64((1 << (14 * limb)) % P), pu)) % pu
        bias = np.array([b % P for b in L["bias"]], dtype=np.uint64)
        out = (out + bias[None, :]) % pu
        act = mulmod_u64(out, out, pu) if li < 2 else out
    return act


def synthetic_images(count, seed=1):
    """MNIST-like sparsity: a pixel is 0 with probability 0.81, else uniform in 1..255"""
    r = np.random.default_rng(seed)
    return np.where(r.random((count, 784)) < 0.81, 0, r.integers(1, 256, size=(count, 784))).astype(float)


def int_logits(w, image, input_scale=16.0):
    """Exact integer model of the CryptoNets / LoLa MNIST network for ONE image (Python integers, no modulus): the wrapper's rounding
    of inputs (`round(pixel / 256 * scale)`), weights and biases, then conv -> square -> dense -> square -> dense.  What a decryption
    must equal modulo the product of the plaintext primes.  `w`: the arrays of CryptoNets/Weights.cs."""
    L = layer_tables(w["Weights_0"], w["Weights_1"], w["Biases_2"], w["Weights_3"], w["Biases_3"])
    act = [int(v) for v in np.rint(np.asarray(image, dtype=np.float64) / 256.0 * input_scale)]
    for li, T in enumerate(L):
        out = [T["bias"][o] + sum(T["W"][o][k] * act[idx] for k, idx in enumerate(T["idx"][o]) if idx >= 0) for o in range(len(T["W"]))]
        act = [v * v for v in out] if li < 2 else out
    return act


def centred(values, M):
    """representatives in (-M/2, M/2] (what DecryptFullPrecision returns for signed vectors)"""
    return [((v % M) - M) if (v % M) * 2 > M else (v % M) for v in values]
