// Internal structures shared by the host runtime and the HIP kernels of libcnhip.so.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define CN_MAXK 12

// modulus with Barrett constant floor(2^128/q) (two words), as SEAL's SmallModulus carries it
struct DMod { uint64_t q, r0, r1; };

// Everything a kernel needs to know about the BFV context; lives in device memory,
// passed by pointer (wave-uniform -> scalar loads).
struct DevConsts {
    uint32_t n, logn, k, kb;
    DMod q[CN_MAXK];
    DMod bsk[CN_MAXK + 1];           // BEHZ base B (k primes) then m_sk
    DMod t;
    // twiddles: for modulus m (0..k-1 = q_j, k..k+kb-1 = bsk_j, k+kb = plain modulus t when batching):
    // tw + m*4*n = {w[n], ws[n], iw[n], iws[n]}
    uint64_t *tw;
    uint64_t ninv[2 * CN_MAXK + 2], ninvs[2 * CN_MAXK + 2];
    uint32_t batching;               // t is prime and == 1 mod 2N: BatchEncoder available
    // exact-FP64 transform path (moduli < 2^49): twd + m*2*n = {w[n], w^-1[n]} as doubles, same bit-reversed order
    double *twd;
    double qd[2 * CN_MAXK + 2], qinvd[2 * CN_MAXK + 2], ninvd[2 * CN_MAXK + 2];
    uint32_t f64ok[2 * CN_MAXK + 2];
    uint32_t q_f64;                  // every coefficient modulus q_j qualifies: ciphertext transforms + key switching run in FP64
    // N = 16384 key switch as two N/2-point transforms per limb (k_keyswitch_split14): sub-transform h of a 2N'-point transform
    // uses root[2m' + h m' + g] where a standalone N'-point transform uses root[m' + g]; twdh + ((j*2 + dir)*2 + h)*N' holds those
    // tables (dir 0 forward, 1 inverse); ninv_w[j] = N^-1 * root^-1[1] closes the last inverse stage
    double *twdh;
    uint64_t ninv_w[CN_MAXK];
    // plaintext scaling (Encryptor::preencrypt / add_plain) and fast plain lift (multiply_plain)
    uint64_t t_half, delta[CN_MAXK], rtq[CN_MAXK], lift_inc[CN_MAXK];
    // BEHZ constants (SEAL util/baseconverter.cpp)
    uint64_t inv_qhat_q[CN_MAXK], mt_inv_qhat_q[CN_MAXK], qhat_mt[CN_MAXK];
    uint64_t qhat_bsk[CN_MAXK + 1][CN_MAXK];
    uint64_t inv_q_mt, q_bsk[CN_MAXK + 1], inv_mt_bsk[CN_MAXK + 1], inv_q_bsk[CN_MAXK + 1];
    uint64_t inv_bhat_b[CN_MAXK], bhat_q[CN_MAXK][CN_MAXK], bhat_msk[CN_MAXK], inv_B_msk, B_q[CN_MAXK];
    uint64_t t_q[CN_MAXK], t_bsk[CN_MAXK + 1];
    // the same BEHZ steps with the constant factors folded so that every base conversion is ONE lazy 128-bit accumulation
    // followed by ONE Barrett reduction (exact modular identities - outputs unchanged):
    uint64_t ex_Q_bsk[CN_MAXK + 1][CN_MAXK];   // (q/q_j mod b) * m~^-1 mod b
    uint64_t ex_R_bsk[CN_MAXK + 1];            // (q mod b) * m~^-1 mod b
    uint64_t fl_c1_q[CN_MAXK];                 // t * (q/q_j)^-1 mod q_j
    uint64_t fl_T_bsk[CN_MAXK + 1];            // t * q^-1 mod b
    uint64_t fl_N_bsk[CN_MAXK + 1][CN_MAXK];   // b - ((q/q_j mod b) * q^-1 mod b)
    uint64_t fl_A_msk[CN_MAXK];                // (B/b_j mod m_sk) * B^-1 mod m_sk
    // FP64 image of the folded BEHZ constants (exact: every value < 2^49), used by k_behz_extend_f64 / k_behz_floor_f64 when all
    // data AND auxiliary primes are below 2^49 (behz_f64 = 1)
    uint32_t behz_f64;
    struct BehzD {
        double mt_inv_qhat_q[CN_MAXK], ex_R_bsk[CN_MAXK + 1], ex_Q_bsk[CN_MAXK + 1][CN_MAXK];
        double fl_c1_q[CN_MAXK], fl_T_bsk[CN_MAXK + 1], fl_N_bsk[CN_MAXK + 1][CN_MAXK];
        double inv_bhat_b[CN_MAXK], fl_A_msk[CN_MAXK], inv_B_msk, bhat_q[CN_MAXK][CN_MAXK], B_q[CN_MAXK];
    } bd;
    // decryption with the {t, gamma} BEHZ rounding (SEAL decryptor.cpp)
    DMod gamma;
    uint64_t tg_q[CN_MAXK], qhat_t[CN_MAXK], qhat_g[CN_MAXK], neg_inv_q_t, neg_inv_q_g, inv_g_t;
    // key switching
    int32_t dbc, gdbc;
    uint32_t rl_dig[CN_MAXK], gk_dig[CN_MAXK], rl_tot, gk_tot;
    // Decomposition convention of the key switch (cn_set_option("ks_xi")).  0 (default): the digits are those of the RAW residue c_l of source limb l
    // and key (l, d) carries 2^(dbc d) s' in limb l only - the CRT-basis form (q/q_l) [(q/q_l)^-1]_{q_l} = delta_{jl} of SURVEY 9.5.  1: the digits are
    // those of xi_l = [c_l (q/q_l)^-1]_{q_l} and key (l, d) carries the RNS image of (q/q_l) 2^(dbc d) s' (the BEHZ paper's xi_q decomposition) - which is zero in
    // every limb but l, where it is (q/q_l mod q_l) 2^(dbc d) s': the same key shape with the scalar (q/q_l mod q_l) moved from the digits into the key.  Both are
    // exact key switches; the keys of one do not work with the digits of the other.  qhat_q[l][j] = (q/q_l) mod q_j (0 unless j == l: only the diagonal is used).
    uint32_t ks_xi;
    uint64_t qhat_q[CN_MAXK][CN_MAXK];
};

// host-side precompute (cn_tables.cpp). tw_host must hold (k+kb+1)*4*n words; index_map (n entries) receives the
// BatchEncoder slot->coefficient map when batching is possible.
int cn_build_consts(DevConsts *c, uint32_t n, const uint64_t *q, uint32_t k, uint64_t t, int dbc, int gdbc,
                    uint64_t *tw_host, uint32_t *index_map, char *err, size_t errlen);
int cn_default_coeff_modulus_impl(uint32_t n, uint64_t *q);
void cn_build_f64_tables(DevConsts *c, const uint64_t *tw_host, double *twd_host);   // twd_host: (k+kb+1)*2*n doubles
void cn_build_half_tables(DevConsts *c, const uint64_t *tw_host, double *twdh_host);  // twdh_host: k*2*2*(n/2) doubles
