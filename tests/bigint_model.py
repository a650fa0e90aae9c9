"""Independent big-integer models of the BFV evaluator's CIPHERTEXT WORDS (test infrastructure, CPU only).

Purpose: the HIP path is held bit-exact to `oracle/seal32_oracle.c`; until now the oracle's words
were checked against independent mathematics only for the NTT and the decryption rounding - every other operation only at the
decrypted-slot level, where a wrong digit order, plain lift or rounding variant that still decrypts correctly would go unnoticed.
This module restates every evaluator operation as plain mathematics over Python integers - polynomial products by Kronecker
substitution or schoolbook, NO number-theoretic transform, no Barrett / Shoup / lazy arithmetic, no code shared with the oracle - and
`tests/test_oracle_words.py` asserts WORD equality with the oracle.  What each model pins (reference call sites in
`HE Wrapper/AtomicSealBfvVector.cs`):

  add_plain / sub_plain   Delta * m with the upper-half increment                                  :1019,1267
  multiply_plain          dense: negacyclic product with SEAL's fast plain lift; constant: lifted scalar (9.3)   :472,645,803,1136
  multiply (BEHZ)         the exact integer characterisation of DESIGN section 4: m~-corrected lift Y of each operand from its q residues,
                          integer negacyclic tensor, floor(t d / q) - beta with beta from the q-side fast conversion, mod q_j (9.4)
                                                                                                                  :461,839
  relinearize             base-2^dbc digits of every limb of c2, low -> high, times the key of (limb, digit), added to (c0, c1) (9.5)
                                                                                                                  :462,840
  apply_galois / rotate   x -> x^elt on both polynomials, key switch of the permuted c1 with the Galois key, NAF steps (9.6)
                                                                                                                  :864,1420
Keys are data here (they are random): the models read the oracle's key words in the documented layout - NTT form, position p holds the
value of the key polynomial at psi^(2 bitrev(p) + 1) with psi the MINIMAL primitive 2N-th root (found here by search) - and compare
key-switch results in that evaluation domain, point by point, with O(N^2) polynomial evaluation instead of a transform.
"""


# ------------------------------------------------------------------ number theory helpers (independent of the oracle)
def is_primitive_2n_root(psi, n, q):
    return pow(psi, n, q) == q - 1


def minimal_primitive_root(n, q):
    """the smallest psi with psi^n = -1 mod q (SEAL try_minimal_primitive_root), by walking the odd powers of any primitive root"""
    assert (q - 1) % (2 * n) == 0
    g = 2
    while True:
        cand = pow(g, (q - 1) // (2 * n), q)
        if is_primitive_2n_root(cand, n, q):
            break
        g += 1
    best, cur, sq = cand, cand, cand * cand % q
    for _ in range(n):                     # the primitive 2n-th roots are cand^(odd)
        cur = cur * sq % q
        if cur < best:
            best = cur
    return best


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def eval_points(n, q):
    """position p of an NTT-form array <-> evaluation point psi^(2 bitrev(p) + 1)"""
    psi = minimal_primitive_root(n, q)
    bits = n.bit_length() - 1
    return [pow(psi, 2 * bitrev(p, bits) + 1, q) for p in range(n)]


def evaluate(poly, points, q):
    """values of the polynomial (list of ints, low -> high) at every point, Horner, O(N^2)"""
    out = []
    for x in points:
        acc = 0
        for c in reversed(poly):
            acc = (acc * x + c) % q
        out.append(acc)
    return out


# ------------------------------------------------------------------ negacyclic products over the integers
def _kron_nonneg(a, b, bits):
    """product of two polynomials with NON-NEGATIVE integer coefficients by Kronecker substitution: evaluate both at 2^bits (coefficients
    packed as fixed-width little-endian fields), ONE big-integer multiplication, read the product's coefficients back field by field"""
    w = (bits + 7) // 8
    A = int.from_bytes(b"".join(int(c).to_bytes(w, "little") for c in a), "little")
    B = int.from_bytes(b"".join(int(c).to_bytes(w, "little") for c in b), "little")
    terms = len(a) + len(b) - 1
    raw = (A * B).to_bytes(w * (terms + 1), "little")
    return [int.from_bytes(raw[w * i:w * (i + 1)], "little") for i in range(terms)]


def negacyclic_mul(a, b):
    """a * b mod (x^N + 1) over Z for integer (possibly negative) coefficient lists of length N"""
    n = len(a)
    assert len(b) == n
    if n <= 64:                                     # schoolbook (the definition)
        out = [0] * n
        for i, x in enumerate(a):
            if x:
                for l, y in enumerate(b):
                    if i + l < n:
                        out[i + l] += x * y
                    else:
                        out[i + l - n] -= x * y
        return out
    ap, an = [max(int(x), 0) for x in a], [max(-int(x), 0) for x in a]
    bp, bn = [max(int(x), 0) for x in b], [max(-int(x), 0) for x in b]
    bits = (max(max(ap), max(an), 1)).bit_length() + (max(max(bp), max(bn), 1)).bit_length() + n.bit_length() + 1
    full = [0] * (2 * n - 1)
    for x, y, s in ((ap, bp, 1), (an, bn, 1), (ap, bn, -1), (an, bp, -1)):
        if any(x) and any(y):
            for i, v in enumerate(_kron_nonneg(x, y, bits)):
                full[i] += s * v
    return [full[i] - (full[i + n] if i + n < 2 * n - 1 else 0) for i in range(n)]


# ------------------------------------------------------------------ ciphertext <-> integer polynomials
def ct_limbs(words, polys, k, n):
    """flat [poly][limb][N] words -> nested python-int lists"""
    w = [int(x) for x in words]
    assert len(w) == polys * k * n
    return [[w[(p * k + j) * n:(p * k + j + 1) * n] for j in range(k)] for p in range(polys)]


def flatten(ct):
    return [x for poly in ct for limb in poly for x in limb]


def product(qs):
    out = 1
    for x in qs:
        out *= x
    return out


# ------------------------------------------------------------------ linear operations
def add_plain(ct, plain, q, t, subtract=False):
    """c0 +- (Delta m + [m >= (t+1)/2] (q mod t)), Delta = floor(q / t), everything reduced per limb; c1 untouched"""
    Q = product(q)
    delta, r, thr = Q // t, Q % t, (t + 1) // 2
    out = [[list(l) for l in p] for p in ct]
    for j, qj in enumerate(q):
        for i, m in enumerate(plain):
            m = int(m)
            s = (delta * m + (r if m >= thr else 0)) % qj
            out[0][j][i] = (out[0][j][i] - s) % qj if subtract else (out[0][j][i] + s) % qj
    return out


def plain_lift(m, qj, t):
    """SEAL's fast plain lift (t < q_j): m stands for the centred representative m - t when m >= (t+1)/2"""
    m = int(m)
    return m + (qj - t) if m >= (t + 1) // 2 else m


def multiply_plain(ct, plain, q, t):
    """every polynomial of the ciphertext times the lifted plaintext, negacyclically.  A plaintext with one non-zero coefficient takes
    SEAL's monomial path - the same mathematical product (negacyclic shift by e, times lift(c)), modelled as such"""
    nz = [i for i, m in enumerate(plain) if int(m)]
    if not nz:
        raise ValueError("plain cannot be zero")
    n = len(ct[0][0])
    out = []
    for poly in ct:
        limbs = []
        for j, qj in enumerate(q):
            lifted = [plain_lift(m, qj, t) if int(m) else 0 for m in plain] + [0] * (n - len(plain))
            limbs.append([v % qj for v in negacyclic_mul(poly[j], lifted)])
        out.append(limbs)
    return out


# ------------------------------------------------------------------ BEHZ multiplication
M_TILDE = 1 << 32


def behz_lift(residues, q):
    """The integer the m~-corrected base extension of SEAL represents, from the q residues c_j of one coefficient:
    X = sum_j [c_j m~ (q/q_j)^-1]_{q_j} (q/q_j)  (fastbconv_mtilde: NOT reduced mod q),  r = [-X q^-1]_{m~} centred (mont_rq),
    Y = (X + q r) / m~  - an exact division; Y == c mod q_j for every j and |Y| <= q (1/2 + k / m~)."""
    Q = product(q)
    X = 0
    for c, qj in zip(residues, q):
        Qj = Q // qj
        X += (int(c) * M_TILDE * pow(Qj, -1, qj) % qj) * Qj
    r = (-X * pow(Q, -1, M_TILDE)) % M_TILDE
    if r >= M_TILDE // 2:
        r -= M_TILDE
    num = X + Q * r
    assert num % M_TILDE == 0
    return num // M_TILDE


def behz_floor(d, q, t):
    """fast_floor + Shenoy-Kumaresan on the integer tensor coefficient d: W = floor(t d / q) - beta, where the fast base conversion of
    the q side, S = sum_j [x_j (q/q_j)^-1]_{q_j} (q/q_j) with x = t d, overshoots x mod q by beta q, beta in [0, k)"""
    Q = product(q)
    x = t * d
    S = 0
    for qj in q:
        Qj = Q // qj
        S += ((x % qj) * pow(Qj, -1, qj) % qj) * Qj
    beta, rem = divmod(S - (x % Q), Q)
    assert rem == 0 and 0 <= beta < len(q)
    return x // Q - beta                                  # python floor division: towards minus infinity


def multiply(a, b, q, t):
    """size-2 x size-2 -> size-3 ciphertext words"""
    k, n = len(q), len(a[0][0])
    lift = lambda ct: [[behz_lift([poly[j][i] for j in range(k)], q) for i in range(n)] for poly in ct]
    A, B = lift(a), lift(b)
    d0 = negacyclic_mul(A[0], B[0])
    d1 = [x + y for x, y in zip(negacyclic_mul(A[0], B[1]), negacyclic_mul(A[1], B[0]))]
    d2 = negacyclic_mul(A[1], B[1])
    out = []
    for d in (d0, d1, d2):
        W = [behz_floor(v, q, t) for v in d]
        out.append([[w % qj for w in W] for qj in q])
    return out


# ------------------------------------------------------------------ key switching, in the evaluation domain
def digits_of(limb, qj, dbc):
    """base-2^dbc digit polynomials of one limb's canonical residues, low -> high; ceil(bits(q_j) / dbc) of them"""
    nd = -(-qj.bit_length() // dbc)
    mask = (1 << dbc) - 1
    return [[(int(c) >> (dbc * d)) & mask for c in limb] for d in range(nd)]


def xi_residues(limb, l, q):
    """xi_l = [c_l (q/q_l)^-1]_{q_l}: the source residues of the BEHZ-paper form of the key switch (ks_xi)"""
    ql = q[l]
    qhat = 1
    for j, qj in enumerate(q):
        if j != l:
            qhat *= qj
    inv = pow(qhat % ql, -1, ql)
    return [(int(c) * inv) % ql for c in limb]


def key_switch_eval(target, key_words, q, dbc, n, xi=False):
    """sum over (source limb l, digit d) of digit polynomial x key (l, d), component c in {0, 1}, as VALUES at the evaluation points
    of every output limb j: acc[c][j][p].  key_words: flat [(l, d)][2][k][N] NTT-form words.  xi: the digits are those of
    [c_l (q/q_l)^-1]_{q_l} instead of c_l (the other self-consistent convention, oracle gen_ksk / libcnhip "ks_xi")."""
    k = len(q)
    kw = [int(x) for x in key_words]
    pts = [eval_points(n, qj) for qj in q]
    acc = [[[0] * n for _ in range(k)] for _ in range(2)]
    pos = 0
    for l in range(k):
        for dig in digits_of(xi_residues(target[l], l, q) if xi else target[l], q[l], dbc):
            for j, qj in enumerate(q):
                vals = evaluate(dig, pts[j], qj)                       # the SAME small integers are residues in every limb
                for c in range(2):
                    base = ((pos * 2 + c) * k + j) * n
                    for p in range(n):
                        acc[c][j][p] = (acc[c][j][p] + vals[p] * kw[base + p]) % qj
            pos += 1
    assert pos * 2 * k * n == len(kw), "key has another digit count than ceil(bits(q_l) / dbc) per limb"
    return acc, pts


def assert_key_switched(out, add0, add1, target, key_words, q, dbc, xi=False):
    """out == (add0 + KS(target)_0, add1 + KS(target)_1), compared through the values of (out - add) at all N evaluation points of
    every limb (N distinct points determine a polynomial of degree < N: this pins every word).  add1 None = zero."""
    k, n = len(q), len(target[0])
    acc, pts = key_switch_eval(target, key_words, q, dbc, n, xi=xi)
    for c, add in ((0, add0), (1, add1)):
        for j, qj in enumerate(q):
            diff = [(int(out[c][j][i]) - (int(add[j][i]) if add is not None else 0)) % qj for i in range(n)]
            assert evaluate(diff, pts[j], qj) == acc[c][j], "key switch differs (component %d, limb %d)" % (c, j)
            assert all(0 <= int(v) < qj for v in out[c][j])


# ------------------------------------------------------------------ Galois automorphisms
def galois_poly(limb, elt, qj):
    """x -> x^elt on one coefficient-form limb mod (x^N + 1)"""
    n = len(limb)
    out = [0] * n
    for i, c in enumerate(limb):
        e = (i * elt) % (2 * n)
        out[e % n] = (qj - int(c)) % qj if e >= n else int(c)
    return out


def galois_elt_from_step(steps, n):
    m = 2 * n
    if steps == 0:
        return m - 1
    s = steps if steps > 0 else n // 2 + steps
    return pow(3, s, m)


def naf(value):
    """non-adjacent form, low-order terms first (SEAL util naf())"""
    res, sign, v, i = [], value < 0, abs(value), 0
    while v:
        zi = (2 - (v & 3)) if v & 1 else 0
        v = (v - zi) >> 1
        if zi:
            res.append((-zi if sign else zi) * (1 << i))
        i += 1
    return res
