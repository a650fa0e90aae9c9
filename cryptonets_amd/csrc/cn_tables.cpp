// Host-side parameter/table precompute for libcnhip.so (runs once per context).
// What SEALContext.Create / SmallNTTTables / BaseConverter::generate compute inside SEAL 3.2
// for `AtomicSealBfvEncryptedEnvironment.GenerateEncryptionKeys` (AtomicSealBfvVector.cs:163-173).
#include "cn_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned __int128 u128;

namespace {

uint64_t mulm(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((u128)a * b % m); }
uint64_t powm(uint64_t b, uint64_t e, uint64_t m) {
    uint64_t r = 1 % m; b %= m;
    for (; e; e >>= 1, b = mulm(b, b, m)) if (e & 1) r = mulm(r, b, m);
    return r;
}
uint64_t invm_prime(uint64_t a, uint64_t p) { return powm(a % p, p - 2, p); }

bool is_prime_u64(uint64_t n) {                      // deterministic Miller-Rabin for 64-bit
    if (n < 2) return false;
    for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        if (n % p == 0) return n == p;
    }
    uint64_t d = n - 1; int s = 0;
    while (!(d & 1)) { d >>= 1; s++; }
    for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        uint64_t x = powm(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < s && comp; i++) { x = mulm(x, x, n); if (x == n - 1) comp = false; }
        if (comp) return false;
    }
    return true;
}

void set_mod(DMod &m, uint64_t q) {
    m.q = q;
    u128 hi = ((u128)1 << 64) / q, rem = ((u128)1 << 64) % q;
    m.r1 = (uint64_t)hi; m.r0 = (uint64_t)((rem << 64) / q);
}

uint32_t brev(uint32_t x, uint32_t bits) { uint32_t r = 0; while (bits--) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

// minimal primitive 2n-th root of unity mod q (what SEAL's SmallNTTTables uses)
bool min_root(uint32_t n, uint64_t q, uint64_t &psi) {
    uint64_t order = 2ull * n;
    if ((q - 1) % order) return false;
    uint64_t g = 0;
    for (uint64_t c = 2; c < 4096 && !g; c++) { uint64_t r = powm(c, (q - 1) / order, q); if (powm(r, n, q) == q - 1) g = r; }
    if (!g) return false;
    uint64_t sq = mulm(g, g, q), cur = g; psi = g;
    for (uint32_t i = 0; i < n; i++, cur = mulm(cur, sq, q)) if (cur < psi) psi = cur;
    return true;
}

bool fill_twiddles(uint64_t *dst, uint32_t n, uint32_t logn, uint64_t q, uint64_t &ninv, uint64_t &ninvs) {
    uint64_t psi;
    if (!min_root(n, q, psi)) return false;
    uint64_t ipsi = invm_prime(psi, q), p = 1, ip = 1;
    uint64_t *w = dst, *ws = dst + n, *iw = dst + 2 * (size_t)n, *iws = dst + 3 * (size_t)n;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t r = brev(i, logn);
        w[r] = p; ws[r] = (uint64_t)(((u128)p << 64) / q);
        iw[r] = ip; iws[r] = (uint64_t)(((u128)ip << 64) / q);
        p = mulm(p, psi, q); ip = mulm(ip, ipsi, q);
    }
    ninv = invm_prime(n, q); ninvs = (uint64_t)(((u128)ninv << 64) / q);
    return true;
}

uint64_t prod_except(const uint64_t *v, uint32_t cnt, int skip, uint64_t m) {
    uint64_t r = 1 % m;
    for (uint32_t i = 0; i < cnt; i++) if ((int)i != skip) r = mulm(r, v[i] % m, m);
    return r;
}
uint32_t ndigits(uint64_t q, int dbc) { uint32_t c = 0; while (q) { c++; q = dbc >= 64 ? 0 : q >> dbc; } return c; }

}  // namespace

int cn_default_coeff_modulus_impl(uint32_t n, uint64_t *q) {
    // DefaultParams.CoeffModulus128(n) of SEAL 3.2 (values verified prime / NTT-friendly in SURVEY 9.1)
    static const uint64_t t2048[] = {0x3fffffff000001ull};
    static const uint64_t t4096[] = {0xffffee001ull, 0xffffc4001ull, 0x1ffffe0001ull};
    static const uint64_t t8192[] = {0x7fffffd8001ull, 0x7fffffc8001ull, 0xfffffffc001ull, 0xffffff6c001ull, 0xfffffebc001ull};
    static const uint64_t t16384[] = {0xfffffffd8001ull, 0xfffffffa0001ull, 0xfffffff00001ull, 0x1fffffff68001ull, 0x1fffffff50001ull,
                                      0x1ffffffee8001ull, 0x1ffffffea0001ull, 0x1ffffffe88001ull, 0x1ffffffe48001ull};
    const uint64_t *tab; int cnt;
    switch (n) {
        case 2048: tab = t2048; cnt = 1; break;
        case 4096: tab = t4096; cnt = 3; break;
        case 8192: tab = t8192; cnt = 5; break;
        case 16384: tab = t16384; cnt = 9; break;
        default: return 0;
    }
    memcpy(q, tab, 8 * cnt);
    return cnt;
}

int cn_build_consts(DevConsts *c, uint32_t n, const uint64_t *q, uint32_t k, uint64_t t, int dbc, int gdbc,
                    uint64_t *tw_host, uint32_t *index_map, char *err, size_t errlen) {
    memset(c, 0, sizeof *c);
    // (the FP64 twiddle image is derived from tw_host by cn_build_f64_tables once tw_host is complete)
    if (k == 0 || k > CN_MAXK) { snprintf(err, errlen, "coeff modulus count %u out of range", k); return -1; }
    if (n < 4 || (n & (n - 1))) { snprintf(err, errlen, "poly modulus degree must be a power of two"); return -1; }
    if (dbc < 1 || dbc > 60 || gdbc < 1 || gdbc > 60) { snprintf(err, errlen, "decomposition bit count must be in [1,60]"); return -1; }
    c->n = n; c->k = k; c->kb = k + 1; c->dbc = dbc; c->gdbc = gdbc;
    while ((1u << c->logn) < n) c->logn++;
    for (uint32_t j = 0; j < k; j++) {
        if (q[j] >> 61 || !is_prime_u64(q[j]) || t >= q[j]) { snprintf(err, errlen, "coeff modulus %u invalid (needs prime < 2^61, > t)", j); return -1; }
        for (uint32_t i = 0; i < j; i++) if (q[i] == q[j]) { snprintf(err, errlen, "coeff moduli must be distinct"); return -1; }
        set_mod(c->q[j], q[j]);
    }
    set_mod(c->t, t);
    // BEHZ auxiliary primes by SEAL's rule: the largest 61-bit primes == 1 (mod 2^18), decreasing:
    // m_sk, gamma (decryption only, unused on the device), then the aux base B.
    std::vector<uint64_t> aux;
    for (uint64_t x = (1ull << 61) - (1ull << 18) + 1; aux.size() < (size_t)k + 2; x -= 1ull << 18) if (is_prime_u64(x)) aux.push_back(x);
    std::vector<uint64_t> bsk(k + 1);
    for (uint32_t i = 0; i < k; i++) bsk[i] = aux[i + 2];
    bsk[k] = aux[0];
    // FP64-friendly auxiliary base.  The words BEHZ multiplication returns do not depend on WHICH primes form Bsk: the m~-corrected
    // extension represents the same integer (c + u q)/m~ in any base, the tensor product is an integer convolution, fast_floor
    // yields floor(t d / q) - beta with beta a function of the q-base residues only, and the Shenoy-Kumaresan conversion back to q
    // is exact whenever |floor(t d / q)| < B m_sk / 2.  SEAL's k+1 primes of 61 bits need integer (Shoup) transforms; k+1 primes just
    // below 2^49 run on the exact-FP64 transform kernels like the data primes do.  Used when every q_j < 2^49 and the base is large
    // enough with 2 bits to spare:  log2(t) + log2(N) + log2(q) + 2 < log2(B m_sk)   (|d| <= N q^2 / 2 (1 + eps)).
    // Nothing in that argument needs B to have exactly k primes either: where k + 1 primes below 2^49 fall a few bits short (the data
    // primes of N = 16384 are 48-49 bits themselves and t, N take 30+ bits more), k + 2 of them are used (kb = k + 2; k >= 6 only - the
    // kernels are instantiated for those shapes).  CN_SEAL_AUX=1 keeps SEAL's base (A/B runs and the parity test that both bases give
    // the oracle's words); CN_AUX_EXTRA=0 forbids the extra prime.
    {
        bool small = !(getenv("CN_SEAL_AUX") && atoi(getenv("CN_SEAL_AUX")));
        const bool extra = !(getenv("CN_AUX_EXTRA") && !atoi(getenv("CN_AUX_EXTRA")));
        long double need = log2l((long double)t) + (long double)c->logn + 2.0L;
        for (uint32_t j = 0; j < k; j++) { if (q[j] >> 49) small = false; need += log2l((long double)q[j]); }
        if (small) {
            std::vector<uint64_t> sm;
            for (uint64_t x = (1ull << 49) - 2ull * n + 1; sm.size() < (size_t)k + 2 && x > (1ull << 48); x -= 2ull * n) {
                bool used = false;
                for (uint32_t j = 0; j < k; j++) if (q[j] == x) used = true;
                if (!used && is_prime_u64(x)) sm.push_back(x);
            }
            for (uint32_t kb = k + 1; kb <= k + 2 && kb <= sm.size(); kb++) {
                if (kb == k + 2 && !(extra && k >= 6 && k <= 9)) break;
                long double have = 0;
                for (uint32_t i = 0; i < kb; i++) have += log2l((long double)sm[i]);
                if (need < have) {
                    bsk.assign(kb, 0);
                    for (uint32_t i = 0; i + 1 < kb; i++) bsk[i] = sm[i + 1];
                    bsk[kb - 1] = sm[0];
                    c->kb = kb;
                    break;
                }
            }
        }
    }
    const uint32_t nb = c->kb - 1;                            // primes of B; bsk[nb] is m_sk
    for (uint32_t i = 0; i <= nb; i++) set_mod(c->bsk[i], bsk[i]);
    // twiddles
    for (uint32_t m = 0; m < k + c->kb; m++) {
        uint64_t mod = m < k ? q[m] : bsk[m - k];
        if (!fill_twiddles(tw_host + (size_t)m * 4 * n, n, c->logn, mod, c->ninv[m], c->ninvs[m])) {
            snprintf(err, errlen, "modulus 0x%llx has no primitive %u-th root of unity", (unsigned long long)mod, 2 * n); return -1;
        }
    }
    // plain modulus tables + BatchEncoder index map (SEAL batchencoder.cpp: generator 3, bit-reversed positions)
    {
        const uint32_t m = k + c->kb;
        c->batching = is_prime_u64(t) && fill_twiddles(tw_host + (size_t)m * 4 * n, n, c->logn, t, c->ninv[m], c->ninvs[m]);
        if (c->batching && index_map) {
            uint64_t mm = 2ull * n, pos = 1;
            for (uint32_t i = 0; i < n / 2; i++) {
                index_map[i] = brev((uint32_t)((pos - 1) >> 1), c->logn);
                index_map[n / 2 + i] = brev((uint32_t)((mm - pos - 1) >> 1), c->logn);
                pos = (pos * 3) & (mm - 1);
            }
        }
    }
    // Delta = floor(q/t) mod q_j, r_t(q) = q mod t  (multi-precision q)
    std::vector<uint64_t> big(1, 1);
    for (uint32_t j = 0; j < k; j++) {
        uint64_t carry = 0;
        for (auto &w : big) { u128 p = (u128)w * q[j] + carry; w = (uint64_t)p; carry = (uint64_t)(p >> 64); }
        if (carry) big.push_back(carry);
    }
    std::vector<uint64_t> quo(big.size()); u128 rem = 0;
    for (int w = (int)big.size() - 1; w >= 0; w--) { u128 cur = (rem << 64) | big[w]; quo[w] = (uint64_t)(cur / t); rem = cur % t; }
    c->t_half = (t + 1) >> 1;
    for (uint32_t j = 0; j < k; j++) {
        u128 d = 0;
        for (int w = (int)quo.size() - 1; w >= 0; w--) d = ((d << 64) | quo[w]) % q[j];
        c->delta[j] = (uint64_t)d; c->rtq[j] = (uint64_t)rem % q[j]; c->lift_inc[j] = q[j] - t;
    }
    // BEHZ tables
    const uint64_t MT = 1ull << 32;
    uint64_t qprod_mt = 1;
    for (uint32_t i = 0; i < k; i++) qprod_mt = (qprod_mt * q[i]) & (MT - 1);
    uint64_t inv = qprod_mt;                                  // Newton inverse mod 2^32
    for (int i = 0; i < 6; i++) inv *= 2 - qprod_mt * inv;
    c->inv_q_mt = inv & (MT - 1);
    for (uint32_t i = 0; i < k; i++) {
        c->inv_qhat_q[i] = invm_prime(prod_except(q, k, (int)i, q[i]), q[i]);
        c->mt_inv_qhat_q[i] = mulm(c->inv_qhat_q[i], MT % q[i], q[i]);
        uint64_t pm = 1;
        for (uint32_t j = 0; j < k; j++) if (j != i) pm = (pm * q[j]) & (MT - 1);
        c->qhat_mt[i] = pm;
        for (uint32_t b = 0; b <= nb; b++) c->qhat_bsk[b][i] = prod_except(q, k, (int)i, bsk[b]);
        c->B_q[i] = prod_except(bsk.data(), nb, -1, q[i]);
        c->t_q[i] = t % q[i];
    }
    for (uint32_t i = 0; i < nb; i++) {
        c->inv_bhat_b[i] = invm_prime(prod_except(bsk.data(), nb, (int)i, bsk[i]), bsk[i]);
        for (uint32_t j = 0; j < k; j++) c->bhat_q[j][i] = prod_except(bsk.data(), nb, (int)i, q[j]);
        c->bhat_msk[i] = prod_except(bsk.data(), nb, (int)i, bsk[nb]);
    }
    for (uint32_t b = 0; b <= nb; b++) {
        c->q_bsk[b] = prod_except(q, k, -1, bsk[b]);
        c->inv_q_bsk[b] = invm_prime(c->q_bsk[b], bsk[b]);
        c->inv_mt_bsk[b] = invm_prime(MT % bsk[b], bsk[b]);
        c->t_bsk[b] = t % bsk[b];
    }
    c->inv_B_msk = invm_prime(prod_except(bsk.data(), nb, -1, bsk[nb]), bsk[nb]);
    {   // decryption constants; gamma = second largest 61-bit prime == 1 mod 2^18 (aux[1])
        const uint64_t g = aux[1];
        set_mod(c->gamma, g);
        for (uint32_t i = 0; i < k; i++) {
            c->tg_q[i] = mulm(t % q[i], g % q[i], q[i]);
            c->qhat_t[i] = prod_except(q, k, (int)i, t);
            c->qhat_g[i] = prod_except(q, k, (int)i, g);
        }
        // t need not be prime for BFV in general, but every plain modulus of the reference is (batching): inverse by Fermat
        uint64_t qt = prod_except(q, k, -1, t), qg = prod_except(q, k, -1, g);
        uint64_t iqt = is_prime_u64(t) ? invm_prime(qt, t) : 0, iqg = invm_prime(qg, g);
        c->neg_inv_q_t = iqt ? t - iqt : 0; c->neg_inv_q_g = g - iqg;
        c->inv_g_t = is_prime_u64(t) ? invm_prime(g % t, t) : 0;
    }
    for (uint32_t i = 0; i < k; i++) c->fl_c1_q[i] = mulm(c->t_q[i], c->inv_qhat_q[i], q[i]);
    for (uint32_t i = 0; i < nb; i++) c->fl_A_msk[i] = mulm(c->bhat_msk[i], c->inv_B_msk, bsk[nb]);
    for (uint32_t b = 0; b <= nb; b++) {
        c->ex_R_bsk[b] = mulm(c->q_bsk[b], c->inv_mt_bsk[b], bsk[b]);
        c->fl_T_bsk[b] = mulm(c->t_bsk[b], c->inv_q_bsk[b], bsk[b]);
        for (uint32_t i = 0; i < k; i++) {
            c->ex_Q_bsk[b][i] = mulm(c->qhat_bsk[b][i], c->inv_mt_bsk[b], bsk[b]);
            uint64_t v = mulm(c->qhat_bsk[b][i], c->inv_q_bsk[b], bsk[b]);
            c->fl_N_bsk[b][i] = v ? bsk[b] - v : 0;
        }
    }
    for (uint32_t j = 0; j < k; j++) {
        c->rl_dig[j] = ndigits(q[j], dbc); c->gk_dig[j] = ndigits(q[j], gdbc);
        c->rl_tot += c->rl_dig[j]; c->gk_tot += c->gk_dig[j];
    }
    c->ks_xi = 0;
    for (uint32_t l = 0; l < k; l++) for (uint32_t j = 0; j < k; j++) c->qhat_q[l][j] = prod_except(q, k, (int)l, q[j]);
    return 0;
}

// FP64 image of the twiddle tables for every modulus below 2^49: twd_host[m][0] = w, [1] = w^-1 (doubles, exact).
void cn_build_f64_tables(DevConsts *c, const uint64_t *tw_host, double *twd_host) {
    const uint32_t n = c->n, nm = c->k + c->kb + 1;
    c->q_f64 = 1;
    for (uint32_t m = 0; m < nm; m++) {
        const uint64_t q = m < c->k ? c->q[m].q : (m < c->k + c->kb ? c->bsk[m - c->k].q : c->t.q);
        const bool usable = (q >> 49) == 0 && (m < c->k + c->kb || c->batching);
        c->f64ok[m] = usable;
        if (m < c->k && !usable) c->q_f64 = 0;
        if (!usable) continue;
        const uint64_t *w = tw_host + (size_t)m * 4 * n, *iw = w + 2 * (size_t)n;
        double *d = twd_host + (size_t)m * 2 * n;
        for (uint32_t i = 0; i < n; i++) { d[i] = (double)w[i]; d[n + i] = (double)iw[i]; }
        c->qd[m] = (double)q; c->qinvd[m] = 1.0 / (double)q; c->ninvd[m] = (double)c->ninv[m];
    }
    // BEHZ element-wise steps in FP64: needs every data and auxiliary modulus on the FP64 path
    c->behz_f64 = 1;
    for (uint32_t m = 0; m < c->k + c->kb; m++) if (!c->f64ok[m]) c->behz_f64 = 0;
    if (c->behz_f64) {
        const uint32_t k = c->k, nb = c->kb - 1;
        for (uint32_t i = 0; i < k; i++) {
            c->bd.mt_inv_qhat_q[i] = (double)c->mt_inv_qhat_q[i]; c->bd.fl_c1_q[i] = (double)c->fl_c1_q[i]; c->bd.B_q[i] = (double)c->B_q[i];
            for (uint32_t j = 0; j < nb; j++) c->bd.bhat_q[i][j] = (double)c->bhat_q[i][j];
        }
        for (uint32_t i = 0; i < nb; i++) { c->bd.inv_bhat_b[i] = (double)c->inv_bhat_b[i]; c->bd.fl_A_msk[i] = (double)c->fl_A_msk[i]; }
        for (uint32_t b = 0; b <= nb; b++) {
            c->bd.ex_R_bsk[b] = (double)c->ex_R_bsk[b]; c->bd.fl_T_bsk[b] = (double)c->fl_T_bsk[b];
            for (uint32_t i = 0; i < k; i++) { c->bd.ex_Q_bsk[b][i] = (double)c->ex_Q_bsk[b][i]; c->bd.fl_N_bsk[b][i] = (double)c->fl_N_bsk[b][i]; }
        }
        c->bd.inv_B_msk = (double)c->inv_B_msk;
    }
}

// half tables of k_keyswitch_split14 (see cn_internal.h); also fills ninv_w
void cn_build_half_tables(DevConsts *c, const uint64_t *tw_host, double *twdh_host) {
    const uint32_t n = c->n, n2 = n / 2;
    for (uint32_t j = 0; j < c->k; j++) {
        const uint64_t *w = tw_host + (size_t)j * 4 * n, *iw = w + 2 * (size_t)n;
        for (int dir = 0; dir < 2; dir++) {
            const uint64_t *src = dir ? iw : w;
            for (uint32_t h = 0; h < 2; h++) {
                double *d = twdh_host + ((size_t)(j * 2 + dir) * 2 + h) * n2;
                d[0] = 0.0;
                for (uint32_t m = 1; m < n2; m <<= 1)
                    for (uint32_t g = 0; g < m; g++) d[m + g] = (double)src[2 * m + h * m + g];
            }
        }
        c->ninv_w[j] = (uint64_t)((u128)c->ninv[j] * iw[1] % c->q[j].q);
    }
}
