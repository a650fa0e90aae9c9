// Shared internals of the translation units the host runtime of libcnhip.so is built from (round 6: cn_api.hip was one 3 200-line unit):
//   cn_api.hip     contexts, streams, options, buffers, graphs, uploads / downloads, encoder, raw transforms, timing, statistics
//   cn_eval.hip    the evaluator: linear operations, scalar GEMM planning, BEHZ multiply, key switching, rotations
//   cn_client.hip  the data owner's side on the device: keys, sampler, keygen, encrypt, decrypt, noise
//   cn_defer.hip   deferred submission of per-ciphertext calls (queue, hazards, flush) and the consumer side of the lock-free submission ring
//   cn_multi.hip   the key broadcast between contexts (RCCL)
// Everything here is internal: the C ABI is include/cnhip.h.  The element-wise kernels (cn_k_elem.hip.h) have internal linkage - every unit that launches one
// carries its own copy.
#pragma once
#include "cn_runtime.h"
#include <thread>
#include "cn_k_elem.hip.h"
#include <algorithm>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define fail cn_fail
struct DOp; struct DeferQueue; struct GemmPlan; struct GemmArith; struct Tab2; struct RotJob; struct BcastNext; struct Slab;

static const RrOps *const rr_ops[3] = {&cn_rr_u64, &cn_rr_f64, &cn_rr_f64l};
static const KsOps *const ks_ops[3] = {&cn_ks_u64, &cn_ks_f64, &cn_ks_f64l};
enum { DOP_GEMM1 = 0, DOP_ADD, DOP_SUB, DOP_ADDPLAIN, DOP_SUBPLAIN, DOP_MULRELIN, DOP_ENCRYPT,
       DOP_COPY, DOP_MULPLAIN, DOP_ROT, DOP_ROTADD, DOP_COLS, DOP_COLSADD, DOP_SUMSLOTS, DOP_TYPES };      // DOP_COPY .. : staged (gather / batched call / scatter) at flush time
struct DOp {
    int type; int32_t level;
    uint64_t *out;                 // output ciphertext (size 2)
    const uint64_t *a, *b;         // operands (ADD/SUB/MULRELIN: ciphertexts; ADDPLAIN/SUBPLAIN: b = plaintext polynomial)
    uint32_t K; size_t terms;      // GEMM1: K (address, weight) pairs from DeferQueue::addr / ::wt [terms ..)
    const uint64_t *bias;          // GEMM1: plaintext polynomial added to the result (an AddPlain folded in at flush time), or null
    uint64_t nonce = 0, item = 0;  // ENCRYPT: the call's seed and the sampler item of this ciphertext (a = plaintext polynomial or null)
    int64_t arg = 0;               // staged kinds: rotation steps (ROT, ROTADD) / slot count (SUMSLOTS); MULPLAIN: b = plaintext polynomial; ROTADD / COLSADD: b = accumulator
    int32_t fold_first = -1; uint32_t fold_count = 0;   // GEMM1: terms [fold_first, +fold_count) of DeferQueue::folds - zero encryptions folded onto this output (cn_defer_flush)
};
struct DeferQueue {
    std::vector<DOp> ops;
    std::vector<uint64_t> addr, wt;
    struct Haz { int32_t w = -1, r = -1, wop = -1; uint32_t readers = 0; int32_t hd = 0; };   // level of the last writer / deepest reader since / index of the writing op / readers since / heavy depth of the value (defer_push)
    // address -> hazard record: open addressing, cleared by bumping the epoch (a dense-layer call touches 845 records)
    struct HazMap {
        struct E { const uint64_t *key = nullptr; uint32_t epoch = 0; Haz v; };
        std::vector<E> tab = std::vector<E>(1 << 12);
        uint32_t epoch = 1; size_t used = 0;
        static size_t hash(const uint64_t *p) { uint64_t x = (uint64_t)p >> 8; x *= 0x9E3779B97F4A7C15ull; return (size_t)(x >> 20); }
        Haz *find(const uint64_t *p) {
            for (size_t i = hash(p) & (tab.size() - 1);; i = (i + 1) & (tab.size() - 1)) {
                if (tab[i].epoch != epoch) return nullptr;
                if (tab[i].key == p) return &tab[i].v;
            }
        }
        Haz &operator[](const uint64_t *p) {
            if (2 * (used + 1) > tab.size()) grow();
            for (size_t i = hash(p) & (tab.size() - 1);; i = (i + 1) & (tab.size() - 1)) {
                if (tab[i].epoch != epoch) { tab[i].key = p; tab[i].epoch = epoch; tab[i].v = Haz(); used++; return tab[i].v; }
                if (tab[i].key == p) return tab[i].v;
            }
        }
        void grow() {
            std::vector<E> old; old.swap(tab);
            tab.assign(old.size() * 2, E()); used = 0;
            for (const E &e : old) if (e.epoch == epoch) (*this)[e.key] = e.v;
        }
        void clear() { used = 0; if (++epoch == 0) { for (E &e : tab) e.epoch = 0; epoch = 1; } }
    } haz;
    int32_t maxlevel = -1;
    std::vector<std::pair<uint64_t *, size_t>> frees;      // arrays released by the caller while calls were pending: back to the pool after the flush
    struct Fold { int32_t enc; uint64_t w; };               // a zero encryption (index of its DOp) folded into a scalar product with weight w (residue mod t)
    std::vector<Fold> folds;
};
// Small arrays (a per-ciphertext caller allocates every Ciphertext on its own: thousands of 640 KiB arrays per layer) are carved out of
// slabs - one hipMalloc per SLAB_PIECES arrays, neighbours in the address space - and only ever travel between the handles and the pool;
// the slabs themselves are released with the context.
static const size_t SLAB_MAX_ITEM = 8u << 20, SLAB_BYTES = 64u << 20;
struct Slab { char *base; size_t bytes; };
static const uint32_t DEFER_STAGED_MAX = 4;       // per-ciphertext callers: calls on up to this many ciphertexts are queued, larger ones run at once
// out[c] = a[c * (a_bcast ? 0 : 1)] * pt[c * pstride]; a_bcast: ONE ciphertext against `count` plaintexts (row-dot batches)
// Dense MultiplyPlain in two launches (k_lift_ntt, k_mul_plain_fused); ranges / zero plaintexts were checked by the caller
// A row-dot batch whose SumAllSlots chain follows: the product kernel leaves sigma_elt(c1) of every product in `out` ([row][k][N], the chain's first scratch array) and takes
// its transformed ciphertext from `ctn` - both inside the scratch arena the caller has sized for the whole call (no ensure_scratch in between: the arena must not move)
struct BcastNext { uint64_t elt; uint64_t *out, *ctn; };
struct GemmArith { bool small, two; uint32_t lazy; int bits; };
// A scalar GEMM is planned once per (gather table, weight matrix): validation, grouping of the outputs that share a gather list,
// weight tiles in the kernel's layout.  A plan can live in HBM (cn_gemm_plan_create: the weights of a layer are uploaded once, every
// inference only launches) or in the per-call scratch (cn_scalar_gemm).
struct GemmPlan {
    uint32_t O = 0, K = 0, Kp = 0, G = 0, M = 0, MT = 0, lazy = 0, max_in = 0;
    bool small = false, two = false, one = false, has_bias = false, mfma = false;
    uint32_t P = 0, mtiles = 0, ksteps = 0;  // matrix-core form: weight digit planes, 32-row output tiles, 32-term steps
    cn_handle bias_pt = 0; uint32_t bias_count = 0;
    uint64_t nnz = 0;                        // non-zero, non-padded terms (statistics)
    std::vector<char> host;                  // [idx | out_idx | bias_idx | weights], each 256 B aligned
    size_t off_oidx = 0, off_bidx = 0, off_w = 0;
    char *dev = nullptr;                     // persistent plans: device copy of `host`
};
// ---- rotations of n ciphertexts by n DIFFERENT step counts as one launch chain (cn_rotate_rows_many; the queued RotateRows calls of one level).
// A single-image network rotates the 13 masked vectors of an Interleave by 13 different amounts, the 5 maps of a Vectorize by 5: one rotation
// is 2 dependent dispatches per hop, and dependent dispatches are what the latency of such a chain is made of.  The hops of a
// rotation (the element of its step count if the key exists, else its NAF terms - the same decomposition rotate_rec walks, so the words are the
// same) are taken in rounds: round r is ONE two-launch key switch over every ciphertext that has an r-th hop, each with its own key and element
// from a table (KsItem); round 0 reads the source and writes the destination, later rounds work on the destination in place.
struct Tab2 { const NTT_GLOBAL uint64_t *src; NTT_GLOBAL uint64_t *dst; };          // one (source, destination) pair of k_copy_tab
struct RotJob { const uint64_t *src; uint64_t *dst; int steps; std::vector<uint64_t> elts; };

// ---- functions defined in one unit and used by the others
int cn_run_ntt(cn_ctx *c, uint64_t *data, uint32_t limbs, uint32_t base_off, uint32_t nmod, int inverse);
void cn_stagger_forget(cn_ctx *ctx);
const NoiseTab &cn_noise_table();          // thresholds of the noise sampler (cn_client.hip)
std::vector<Slab> &slabs_of(cn_ctx *ctx);
bool in_slab(cn_ctx *ctx, const void *p);
int use(cn_ctx *c);
int ensure_scratch(cn_ctx *c, size_t bytes);
size_t al(size_t b);
char *pin_block(cn_ctx *c, size_t bytes);
int upload_bytes(cn_ctx *c, const void *host, size_t bytes, void *dev);
int place_table(cn_ctx *c, const void *host, size_t bytes, void *fallback, const void **dev);
Buffer *getbuf(cn_ctx *c, cn_handle h, int kind);
int range_ok(const Buffer *b, uint32_t first, uint32_t count, uint32_t stride = 1);
bool streams_share_a_queue(hipStream_t a, hipStream_t b);
int pick_stream(cn_ctx *c);
int ctx_init(cn_ctx *c, uint32_t n, uint32_t k, int device, std::vector<uint64_t> &tw);
void ctx_teardown(cn_ctx *ctx);
bool keys_as_f64(const cn_ctx *ctx);
int set_key(cn_ctx *ctx, KsKey &slot, const uint64_t *words, size_t count, size_t expect, int is_dev, bool coeff_form = false);
void pool_flush(cn_ctx *ctx);
int dev_alloc(cn_ctx *ctx, size_t bytes, uint64_t **out);
int dev_release(cn_ctx *ctx, uint64_t *p, size_t bytes);
int alloc_buf(cn_ctx *ctx, int kind, uint32_t count, uint32_t size, cn_handle *out);
bool submit_async(const cn_ctx *ctx);
int free_body(cn_ctx *ctx, cn_handle h);
int free_many_body(cn_ctx *ctx, const cn_handle *h, uint32_t n);
int free_graph(cn_ctx *ctx, Buffer &b);
int ensure_index_map(cn_ctx *ctx);
int addsub(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op);
int addsub_body(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op);
int add_plain_body(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count);
int mul_plain_fused(cn_ctx *ctx, Buffer *A, uint32_t ai, bool a_bcast, Buffer *P, uint32_t pi, uint32_t pstride, Buffer *O, uint32_t oi, uint32_t count, const BcastNext *nx = nullptr);
bool mul_plain_takes_bcast(cn_ctx *ctx, uint32_t count);
int mul_plain_impl(cn_ctx *ctx, Buffer *A, uint32_t ai, bool a_bcast, Buffer *P, uint32_t pi, uint32_t pstride, Buffer *O, uint32_t oi, uint32_t count, const BcastNext *nx = nullptr);
uint64_t lift_scalar(const DevConsts &hc, uint64_t w, uint32_t j);
bool gemm_weights_small(cn_ctx *ctx, const uint64_t *W, size_t count);
GemmArith gemm_arith(cn_ctx *ctx, bool weights_small);
bool gemm_mfma_ok(cn_ctx *ctx, const GemmArith &ar, uint32_t M, uint32_t K);
uint32_t gemm_weight_planes(cn_ctx *ctx, const uint64_t *W, size_t count);
int free_gemm_plan(cn_ctx *ctx, Buffer &b);
bool pair_gather_lists(uint32_t O, uint32_t &K, std::vector<int32_t> &gidx, const uint64_t *W, std::vector<uint64_t> &W2);
int build_gemm_plan(cn_ctx *ctx, const int32_t *idx, const uint64_t *W, uint32_t O, uint32_t K, Buffer *BP, cn_handle bias_pt, const int32_t *bias_idx, GemmPlan &P);
int run_gemm_plan(cn_ctx *ctx, const GemmPlan &P, const char *tables, Buffer *I, Buffer *OB, uint32_t oi);
bool run_intt_tensor(cn_ctx *c, const uint64_t *A, const uint64_t *B, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm);
bool square_fused_ok(cn_ctx *c, uint32_t base_off, uint32_t Lm, bool &light);
void run_square_fused(cn_ctx *c, const uint64_t *A, size_t astride, const uint64_t *const *atab, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm, bool light);
bool aux_stream_ready(cn_ctx *ctx);
// "sq_halves": Multiply + Relinearize of a batch software-pipelined in parts over the context's two streams (round 6): part i on stream i % 2, its Multiply behind the Multiply
// of the part before it (event), its key switch behind its own Multiply (stream order) - [mul 0][ks 0 | mul 1][ks 1 | mul 2][ks 2]: the HBM-bound base extension / floor and
// the transform kernels of a Multiply fill what the FP64-bound key switch of the part before leaves.  The context's stream continues behind the last part of either stream.
// Three parts of 30 / 40 / 30 % measured best for the CryptoNets batch (profiles/r06_mulrelin_parts.txt: plain loop of the two primes 12.6 -> 12.0-12.2 ms, the half-batch
// stagger of the primes 12.2-12.6; two parts 12.4-12.5, four 12.7, five 12.4); CN_SQ_PARTS / CN_SQ_SPLIT (cut points in per mille) for experiments.
// mul(first, count), ks(first, count) launch on ctx->stream.
static const uint32_t SQ_HALVES_MIN = 512;       // (the 100-ciphertext layer of CryptoNets pipelined as well: 12.35 -> 13.4 ms per batch, visit AY)
template <class FM, class FK> static int pipelined_halves(cn_ctx *ctx, uint32_t c, FM mul, FK ks) {
    static const uint32_t parts_env = [] { const char *e = getenv("CN_SQ_PARTS"); const int v = e ? atoi(e) : 3; return (uint32_t)(v >= 2 && v <= 8 ? v : 3); }();
    const uint32_t P = std::min<uint32_t>(parts_env, c / 128 ? c / 128 : 1);
    if (P < 2) { CHECK(mul(0u, c)); return ks(0u, c); }
    uint32_t first[9]; first[0] = 0;
    for (uint32_t i = 1; i < P; i++) first[i] = (uint32_t)(((uint64_t)c * i / P + 7) & ~7ull);
    first[P] = c;
    static const std::vector<uint32_t> cuts = [] {                    // experiment: CN_SQ_SPLIT="250,625" = the cut points in per mille (P - 1 of them, increasing)
        std::vector<uint32_t> v; const char *e = getenv("CN_SQ_SPLIT");
        while (e && *e) { v.push_back((uint32_t)atoi(e)); e = strchr(e, ','); if (e) e++; }
        return v; }();
    if (cuts.empty() && P == 3) { first[1] = (uint32_t)(((uint64_t)c * 3 / 10 + 7) & ~7ull); first[2] = (uint32_t)(((uint64_t)c * 7 / 10 + 7) & ~7ull); }
    if (cuts.size() + 1 == P) for (uint32_t i = 1; i < P; i++) first[i] = std::min<uint32_t>(c, (uint32_t)(((uint64_t)c * cuts[i - 1] / 1000 + 7) & ~7ull));
    int rc = 0;
    for (uint32_t i = 0; i < P && !rc; i++) {
        const bool aux = (i & 1) != 0;
        if (i) {                                                      // this part's Multiply behind the previous part's (recorded on the other stream)
            if (hipStreamWaitEvent(aux ? ctx->stream2 : ctx->stream, ctx->ev_fork, 0) != hipSuccess) { rc = fail(CN_ERR_HIP, "hipStreamWaitEvent failed"); break; }
        }
        if (aux) std::swap(ctx->stream, ctx->stream2);
        rc = mul(first[i], first[i + 1] - first[i]);
        if (!rc && i + 1 < P && hipEventRecord(ctx->ev_fork, ctx->stream) != hipSuccess) rc = fail(CN_ERR_HIP, "hipEventRecord failed");
        if (!rc) rc = ks(first[i], first[i + 1] - first[i]);
        if (!rc && aux && i + 2 >= P && hipEventRecord(ctx->ev_join, ctx->stream) != hipSuccess) rc = fail(CN_ERR_HIP, "hipEventRecord failed");      // the last part on the second stream
        if (aux) std::swap(ctx->stream, ctx->stream2);
    }
    CHECK(rc);
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return 0;
}
size_t mul_scratch_per_ct(cn_ctx *c, bool square);
int do_multiply(cn_ctx *ctx, const uint64_t *a, uint32_t astride, const uint64_t *b, uint32_t bstride, uint64_t *out3, uint32_t cnt, const uint64_t *const *atab = nullptr, const uint64_t *const *btab = nullptr);
int ensure_ks_part(cn_ctx *ctx, size_t need);
uint32_t ks_digit_max_blocks();
uint32_t ks_wide_max_blocks();
int ks_planned_mode(cn_ctx *ctx, uint32_t cnt, int galois);
int do_keyswitch(cn_ctx *ctx, const uint64_t *target, size_t tstride, const uint64_t *add0, const uint64_t *add1, size_t astride, const KsKey &key, uint64_t *out, uint32_t cnt, int galois, const uint64_t *extra = nullptr, size_t xstride = 0, uint64_t *const *out_tab = nullptr, uint32_t perm_elt = 0, const KsItem *items = nullptr, uint32_t next_elt = 0, uint64_t *next_out = nullptr);
bool ks_pair14_ok(cn_ctx *ctx, uint32_t cnt, int galois, const KsKey &key);
uint32_t chunk_for(cn_ctx *ctx, size_t per_ct, uint32_t count);
int mul_relin_body(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out, uint32_t oi, uint32_t count);
int do_galois(cn_ctx *ctx, const uint64_t *in, uint64_t elt, uint64_t *out, uint64_t *tmp, uint32_t count, const uint64_t *acc = nullptr, const uint64_t *pre = nullptr, uint64_t next_elt = 0, uint64_t *next_out = nullptr);
bool shifted_overlap(const Buffer *I, uint32_t ii, const Buffer *O, uint32_t oi, uint32_t count);
int galois_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, uint64_t elt, Buffer *O, uint32_t oi, uint32_t count);
bool galois_key_present(cn_ctx *ctx, uint64_t elt);
bool has_direct_key(cn_ctx *ctx, int steps);
int rotate_rec(cn_ctx *ctx, uint64_t *cur, int steps, uint64_t *tmp, uint32_t count);
int rotate_rows_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, int steps, Buffer *O, uint32_t oi, uint32_t count);
int rotate_check(cn_ctx *ctx, int steps);
int rotation_hops(cn_ctx *ctx, int steps, std::vector<uint64_t> &elts);
int rotate_jobs(cn_ctx *ctx, std::vector<RotJob> &jobs);
int rotate_rows_add_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, int steps, Buffer *A, uint32_t ai, Buffer *O, uint32_t oi, uint32_t count);
int rotate_columns_add_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, Buffer *A, uint32_t ai, Buffer *O, uint32_t oi, uint32_t count);
std::vector<uint64_t> sum_slots_chain_elts(cn_ctx *ctx, uint32_t count, uint32_t length);
int sum_slots_impl(cn_ctx *ctx, Buffer *H, uint32_t first, uint32_t count, uint32_t length, bool first_ready = false);
int set_plain_key(cn_ctx *ctx, uint64_t **slot, const uint64_t *words, size_t count, size_t expect, bool is_dev = false, bool coeff_form = false);
RngKey rng_key_of(const cn_ctx *ctx);
int sample_poly(cn_ctx *ctx, uint64_t *dst, uint32_t polys, int kind, uint64_t seed, uint64_t stream);
int gen_ksk(cn_ctx *ctx, const uint64_t *snew, int dbc, const uint32_t *dig, uint32_t tot, uint64_t seed, uint64_t *key, uint64_t *e);
int adopt_ksk(cn_ctx *ctx, KsKey &slot, uint64_t *dev, size_t words);
int encrypt_chain(cn_ctx *ctx, uint32_t cnt, const uint64_t *ptd, uint32_t pt_stride_words, uint64_t *out, uint64_t seed, const EncTab *htab);
int encrypt_body(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint32_t pt_stride, cn_handle out, uint32_t oi, uint32_t count, uint64_t seed);
int decrypt_phase(cn_ctx *ctx, Buffer *I, uint32_t ci, uint32_t count, uint64_t *&acc);
DeferQueue *cn_defer_new();
void cn_defer_delete(DeferQueue *q);
bool cn_defer_pending(cn_ctx *ctx);
int32_t defer_level(DeferQueue *q, const uint64_t *const *ins, uint32_t nin, const uint64_t *out);
int defer_push(cn_ctx *ctx, DOp op, const uint64_t *const *ins, uint32_t nin);
int flush_gemm_group(cn_ctx *ctx, DeferQueue *q, const std::vector<const DOp *> &ops, uint32_t K);
int flush_elementwise_group(cn_ctx *ctx, const std::vector<const DOp *> &ops, int type);
int flush_mulrelin_group(cn_ctx *ctx, const std::vector<const DOp *> &all);
int ensure_stage(cn_ctx *ctx, size_t bytes);
int copy_by_table(cn_ctx *ctx, const std::vector<Tab2> &tab, Tab2 *dtab, uint32_t words_per_item);
int flush_staged_group(cn_ctx *ctx, const std::vector<const DOp *> &all, int type);
int defer_staged(cn_ctx *ctx, int type, Buffer *A, uint32_t ai, Buffer *B, uint32_t bi, const uint64_t *plain, uint32_t pstride_words, Buffer *O, uint32_t oi, uint32_t count, int64_t arg);
int flush_encrypt_group(cn_ctx *ctx, const std::vector<const DOp *> &ops);
bool zero_fold_ok(cn_ctx *ctx);
int flush_zero_folds(cn_ctx *ctx, DeferQueue *q, const std::vector<const DOp *> &gemms);
int defer_encrypt(cn_ctx *ctx, const uint64_t *ptd, uint32_t pt_stride_words, Buffer *O, uint32_t oi, uint32_t count, uint64_t seed);
int cn_defer_flush(cn_ctx *ctx);
bool deferring(cn_ctx *ctx);
int scalar_dot_body(cn_ctx *ctx, const cn_handle *in, const uint32_t *in_idx, const uint64_t *w, uint32_t K, cn_handle out, uint32_t oi);
int defer_addsub(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op);
int defer_add_plain(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count);
int defer_mul_relin(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out, uint32_t oi, uint32_t count);
void ready_refill(cn_ctx *ctx);
int ring_exec(cn_ctx *ctx, const SubRec &r);
void ring_drain(cn_ctx *ctx, uint64_t upto);
int ring_sync(cn_ctx *ctx, bool report);
int ring_push(cn_ctx *ctx, uint32_t type, uint32_t count, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t x, uint64_t arg);
int raw_ntt(cn_ctx *ctx, void *p, uint32_t limbs, int base, int inverse);

#define GETCT(var, h, sz) Buffer *var = getbuf(ctx, h, 0); if (!var) return fail(CN_ERR_ARG, "invalid ciphertext handle " #h); \
    if ((sz) && var->size != (uint32_t)(sz)) return fail(CN_ERR_ARG, "ciphertext size mismatch for " #h)
#define GETPT(var, h) Buffer *var = getbuf(ctx, h, 1); if (!var) return fail(CN_ERR_ARG, "invalid plaintext handle " #h)
#define LOCK_ONLY CHECK(use(ctx)); CHECK(ring_sync(ctx, false))
#define API_BODY return ctx->mu.run([&]() -> int {
#define API_END });
#define LOCK CHECK(use(ctx)); CHECK(ring_sync(ctx, true)); CHECK(cn_defer_flush(ctx))

#define launch_count cn_launch_count
#define NOT_CAPTURING(what) do { if (ctx->capturing) return fail(CN_ERR_ARG, what " is not possible while a graph is recorded (cn_graph_begin .. cn_graph_end)"); } while (0)
#define KS_DIGIT_MAX_BLOCKS ks_digit_max_blocks()
#define KS_WIDE_MAX_BLOCKS ks_wide_max_blocks()
#define DISPATCH_K2(fn, ...) switch (ctx->hc.k) { \
    case 1: fn<1>(__VA_ARGS__); break; case 2: fn<2>(__VA_ARGS__); break; case 3: fn<3>(__VA_ARGS__); break; case 4: fn<4>(__VA_ARGS__); break; \
    case 5: fn<5>(__VA_ARGS__); break; case 6: fn<6>(__VA_ARGS__); break; case 7: fn<7>(__VA_ARGS__); break; case 8: fn<8>(__VA_ARGS__); break; \
    case 9: fn<9>(__VA_ARGS__); break; default: return fail(CN_ERR_ARG, "at most 9 coefficient moduli"); }

template <class T> static T *salloc(cn_ctx *c, size_t count) {
    size_t b = al(count * sizeof(T));
    if (c->soff + b > c->scap) return nullptr;
    T *p = (T *)(c->scratch + c->soff); c->soff += b; return p;
}
template <class T> static int upload_tmp(cn_ctx *c, const T *host, size_t count, T **dev) {
    *dev = salloc<T>(c, count);
    if (!*dev) return fail(CN_ERR_HIP, "internal: scratch exhausted");
    return upload_bytes(c, host, count * sizeof(T), *dev);
}

template <class K> static int big_lds(K kern, size_t bytes) {
    HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
template <int EPT> static int set_ks_attr(size_t bytes) {
    HIPCHK(hipFuncSetAttribute((const void *)k_keyswitch<EPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// HOT LOOP A
// weight tiles of a planned GEMM in kernel layout; row(g, m): the K weights (residues mod t) of member m of group g, or null
// tap_ok(g, kk): term kk of group g gathers a ciphertext; a padded tap gets the weight 0 whatever the caller passed (k_scalar_gemm_f64 multiplies a valid word
// by it instead of selecting per lane)
template <class ROW, class TAP> static void pack_gemm_weights(cn_ctx *ctx, uint32_t G, uint32_t M, uint32_t K, bool small, ROW row, TAP tap_ok, uint32_t &MT, std::vector<char> &wbytes) {
    const uint32_t k = ctx->hc.k; const uint64_t t = ctx->hc.t.q;
    if (small) {
        const uint32_t MTf = M >= 16 ? 20 : (M >= 8 ? 10 : (M >= 3 ? 5 : 1)), mtf = (M + MTf - 1) / MTf;
        const uint32_t Kw = gemm_f64_rows(K);
        std::vector<double> hWd((size_t)G * mtf * Kw * MTf, 0.0);           // [g][mtile][kk < Kw][m], zero padded (gemm_f64_rows)
        for (uint32_t g = 0; g < G; g++) for (uint32_t m = 0; m < M; m++) {
            const uint64_t *wr = row(g, m);
            if (!wr) continue;
            double *dst = &hWd[(((size_t)g * mtf + m / MTf) * Kw) * MTf + m % MTf];
            for (uint32_t kk = 0; kk < K; kk++) { uint64_t w = tap_ok(g, kk) ? wr[kk] : 0; dst[(size_t)kk * MTf] = w >= ctx->hc.t_half ? -(double)(t - w) : (double)w; }
        }
        MT = MTf;
        wbytes.assign((const char *)hWd.data(), (const char *)(hWd.data() + hWd.size()));
    } else {
        const uint32_t MTi = M >= 8 ? 10 : (M >= 3 ? 5 : 1), mti = (M + MTi - 1) / MTi;
        std::vector<uint64_t> hW((size_t)k * G * mti * K * MTi, 0);         // [j][g][mtile][kk][m], zero padded
        for (uint32_t j = 0; j < k; j++) for (uint32_t g = 0; g < G; g++) for (uint32_t m = 0; m < M; m++) {
            const uint64_t *wr = row(g, m);
            if (!wr) continue;
            uint64_t *dst = &hW[((((size_t)j * G + g) * mti + m / MTi) * K) * MTi + m % MTi];
            for (uint32_t kk = 0; kk < K; kk++) { uint64_t w = tap_ok(g, kk) ? wr[kk] : 0; dst[(size_t)kk * MTi] = w ? lift_scalar(ctx->hc, w, j) : 0; }
        }
        MT = MTi;
        wbytes.assign((const char *)hW.data(), (const char *)(hW.data() + hW.size()));
    }
}

// One-limb form of the small-weight kernel: sum_k w_k x_k with x_k < q_max is an exact double as long as (sum_k |w_k| + 1) q_max <= 2^53 for every output row
// (all partial sums are integers below 2^53; the + 1 leaves room for the recentred carry of the fold) - the words are then not split into limbs at all: ONE FMA
// per MAC instead of two, no masks and shifts, one recentring per output.  True for the CryptoNets convolution (row sums <= 373 with the trained weights,
// 44-bit moduli); the dense layers have larger row sums and keep the two-limb form (or the matrix cores).  Needs the whole term list in one block (K <= lazy).
template <class ROW, class TAP> static bool gemm_one_limb(cn_ctx *ctx, const GemmArith &ar, uint32_t G, uint32_t M, uint32_t K, ROW row, TAP tap_ok) {
    static const bool on = !(getenv("CN_GEMM_ONE_LIMB") && !atoi(getenv("CN_GEMM_ONE_LIMB")));
    if (!on || !ar.small || ar.bits > 49 || K > ar.lazy) return false;
    uint64_t qmax = 0; for (uint32_t j = 0; j < ctx->hc.k; j++) qmax = std::max(qmax, ctx->hc.q[j].q);
    const uint64_t room = (1ull << 53) / qmax;                 // sum |w| + 1 <= room
    const uint64_t t = ctx->hc.t.q;
    for (uint32_t g = 0; g < G; g++) for (uint32_t m = 0; m < M; m++) {
        const uint64_t *wr = row(g, m);
        if (!wr) continue;
        uint64_t sum = 1;
        for (uint32_t kk = 0; kk < K; kk++) if (tap_ok(g, kk)) { const uint64_t w = wr[kk]; sum += w >= ctx->hc.t_half ? t - w : w; if (sum > room) return false; }
    }
    return true;
}
// fragments [g][p][mtile][kstep][lane][16]: lane l, byte t = digit p of the weight of output row 32 mtile + (l & 31) for term
// 32 kstep + 16 (l >> 5) + t; zero for padded rows / terms / taps.  row(g, m): residues mod t of member m, or null; tap_ok(g, kk).
template <class ROW, class TAP> static void pack_gemm_mfma(cn_ctx *ctx, uint32_t G, uint32_t M, uint32_t K, uint32_t P, ROW row, TAP tap_ok, std::vector<char> &wbytes) {
    const uint32_t mtiles = (M + 31) / 32, ksteps = (K + 31) / 32;
    const uint64_t t = ctx->hc.t.q;
    wbytes.assign((size_t)G * P * mtiles * ksteps * 1024, 0);
    for (uint32_t g = 0; g < G; g++) for (uint32_t m = 0; m < M; m++) {
        const uint64_t *wr = row(g, m);
        if (!wr) continue;
        const uint32_t mt = m / 32, r = m % 32;
        for (uint32_t kk = 0; kk < K; kk++) {
            const uint64_t w = wr[kk];
            if (!w || !tap_ok(g, kk)) continue;
            const int64_t sw = w >= ctx->hc.t_half ? -(int64_t)(t - w) : (int64_t)w;
            const uint32_t rec = ((uint32_t)(sw + 0x808080) ^ 0x808080u);          // byte p = signed digit p
            const uint32_t ks = kk / 32, half = (kk % 32) / 16, tt = kk % 16;
            for (uint32_t p = 0; p < P; p++)
                wbytes[((((size_t)g * P + p) * mtiles + mt) * ksteps + ks) * 1024 + (size_t)(half * 32 + r) * 16 + tt] = (char)(uint8_t)(rec >> (8 * p));
        }
    }
}
