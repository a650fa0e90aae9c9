// Host-side structures of libcnhip.so shared by the runtime units (cn_api.hip, cn_eval.hip, cn_client.hip, cn_defer.hip, cn_multi.hip: cn_api_shared.h) and the kernel-launch translation units
// (cn_l_*.hip).  The library is built from several translation units so that hipcc compiles the kernel families in parallel: the
// register-radix kernels alone are ~180 instantiations (5 transform sizes x 3 arithmetic policies x 12 kernels).
#pragma once
#include "../../include/cnhip.h"
#include "cn_internal.h"
#include "cn_submit.h"
#include <hip/hip_runtime.h>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

int cn_fail(int code, const char *fmt, ...);       // sets the thread-local message of cn_last_error(), returns `code`
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return cn_fail(CN_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); } while (0)
#define CHECK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

struct GemmPlan;
// A captured operation sequence (cn_graph_begin / cn_graph_end): the instantiated HIP graph, the host blocks its upload nodes read at
// every launch, and the device arrays that were handed out while it was recorded and are not owned by a live handle - they stay
// reserved for the graph (its kernels carry their addresses), out of the pool, until the graph is freed.
struct CapturedGraph {
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    std::vector<std::unique_ptr<char[]>> staged;
    std::vector<std::pair<uint64_t *, size_t>> reserved;
};
struct Buffer {
    int kind;                 // 0 = ciphertext array, 1 = dense plaintext array, 2 = scalar GEMM plan, 3 = captured graph
    std::shared_ptr<GemmPlan> plan;
    std::shared_ptr<CapturedGraph> cg;
    uint32_t count, size;     // size = polys per ciphertext
    uint64_t *d;
    size_t item_words;
    std::vector<uint8_t> pt_zero;   // per plaintext: all coefficients zero?
};
// Handle table: a handle is (generation << 32) | (slot + 1) - looked up by indexing, not hashing (a dense-layer call of the unchanged
// caller names 845 input handles), stale handles are recognised by their generation.
class HandleTable {
    struct Slot { Buffer b; uint32_t gen = 0; bool live = false; };
    std::vector<Slot> slots;
    std::vector<uint32_t> free_;
    size_t live_ = 0;
public:
    Buffer *find(cn_handle h) {
        const uint32_t s = (uint32_t)h - 1u;
        if ((uint32_t)h == 0 || s >= slots.size() || !slots[s].live || slots[s].gen != (uint32_t)(h >> 32)) return nullptr;
        return &slots[s].b;
    }
    cn_handle insert(Buffer &&b) {
        uint32_t s;
        if (!free_.empty()) { s = free_.back(); free_.pop_back(); } else { s = (uint32_t)slots.size(); slots.emplace_back(); }
        slots[s].b = std::move(b); slots[s].live = true; slots[s].gen = (slots[s].gen + 1) & 0x7fffffffu;
        live_++;
        return ((cn_handle)slots[s].gen << 32) | (cn_handle)(s + 1);
    }
    void erase(cn_handle h) {
        const uint32_t s = (uint32_t)h - 1u;
        slots[s].live = false; slots[s].b = Buffer(); free_.push_back(s); live_--;
    }
    size_t size() const { return live_; }
    template <class F> void for_each(F f) { for (Slot &s : slots) if (s.live) f(s.b); }
};
struct KsKey { uint64_t *d; bool owned; bool f64; };   // f64: words converted to doubles for the FP64 key-switch kernel

// The contexts are called from Defaults.ThreadCount threads at once (HE Wrapper/Utils.cs:46-88) and the critical sections are a few
// hundred nanoseconds of bookkeeping (handle table, deferred-operation queue): a futex mutex hands every contended acquisition through
// the kernel, so waiters spin (and yield when the wait gets long).
// Spin lock (cn_host.cpp): the critical sections are short, waiters spin briefly and then yield.
// A request published by a thread that found the lock held: the holder executes it (combining) and reports back through it.
struct CnReq {
    int (*fn)(void *) = nullptr; void *arg = nullptr; int rc = 0;
    std::atomic<int> done{0}, asleep{0};
    CnReq *next = nullptr;
    char err[256];
};
class CnMutex {
public:
    struct Node {};           // (per-acquisition state of a queue lock; the current lock needs none)
    void lock(Node &n);
    void unlock(Node &n);
    // Run f() under the lock.  Default: plain acquisition.  With CN_LOCK_COMBINE=1 (an experiment that is kept switchable, measured slower -
    // cn_host.cpp): on this thread when the lock is free - and then ALSO every request other threads published meanwhile - or, when it is
    // held, on the holder's thread (the request is published, this thread waits for its completion), so that the few cache lines every call
    // works on (queue tails, hazard table, handle table, counters) stay in one core's cache.  f must not call into the same context again.
    template <class F> int run(F &&f) {
        if (!combining()) { Node n; lock(n); const int rc = f(); unlock(n); return rc; }
        if (try_take_me()) { const int rc = f(); serve(); release(); return rc; }
        typedef typename std::remove_reference<F>::type Fn;
        CnReq r;
        r.fn = [](void *p) -> int { return (*static_cast<Fn *>(p))(); };
        r.arg = (void *)&f; r.err[0] = 0;
        return submit(r);
    }
    // the lock-free submission path (cn_submit.h): whoever finds the lock free drains the ring; nobody waits
    bool try_lock() { return try_take_me(); }
    void unlock_now() { release(); }
    bool is_held() const { return held.load(std::memory_order_relaxed) != 0; }
private:
    std::atomic<int> held{0};
    std::atomic<int> spinners{0};      // waiters that spin; the others sleep on wake_seq (cn_host.cpp)
    std::atomic<int> sleepers{0}, wake_seq{0};
    std::atomic<uint32_t> last_owner{0}, burst{0};   // owner bias (cn_host.cpp): who released last, how many times in a row it re-acquired
    std::atomic<uint64_t> released_at{0};            // TSC of the last release
    std::atomic<CnReq *> pending{nullptr};           // published requests (a stack; served oldest first)
    bool try_take(uint32_t me);
    bool try_take_me();
    static bool combining();
    void release();
    void serve();
    int submit(CnReq &r);
};
struct CnGuard {
    CnMutex &m; CnMutex::Node n;
    explicit CnGuard(CnMutex &m_) : m(m_) { m.lock(n); }
    ~CnGuard() { m.unlock(n); }
    CnGuard(const CnGuard &) = delete;
    CnGuard &operator=(const CnGuard &) = delete;
};

struct DeferQueue;            // cn_api_shared.h / cn_defer.hip
struct cn_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DevConsts hc;             // host copy
    DevConsts *dc = nullptr;  // device copy
    uint64_t *tw = nullptr;
    CnMutex mu;
    HandleTable bufs;
    KsKey rlk{nullptr, false, false};
    double *twd = nullptr, *twdh = nullptr;
    bool use_f64 = true;      // CN_NO_F64=1 / cn_set_option("f64",0): integer (Shoup) transforms everywhere
    std::map<uint64_t, KsKey> gk;
    uint64_t *sk = nullptr, *pk = nullptr;   // client-side keys (NTT form) when the data owner's GPU runs keygen/encrypt/decrypt
    uint64_t rng_item = 0;                    // running polynomial counter of the sampler (part of the ChaCha20 block counter)
    uint32_t rng_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 256-bit ChaCha20 key of the sampler (cn_set_rng_key; cn_set_rng_salt sets the first 64 bits)
    char *scratch = nullptr; size_t scap = 0, soff = 0, smax;
    std::vector<std::unique_ptr<char[]>> staged;   // host blocks of in-flight upload_tmp copies
    cn_stats st{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_order = nullptr;     // cn_ctx_wait_for: marks a point of this context's stream another context waits for
    uint32_t bs, chunks;      // element-wise geometry
    std::vector<uint32_t> index_map;   // BatchEncoder slot -> coefficient position
    uint32_t *d_index_map = nullptr;   // the same table in HBM (cn_encode_batch / cn_decode_batch)
    size_t ctw2;              // words of a size-2 ciphertext
    bool legacy_ntt = false;  // CN_LEGACY_NTT=1: radix-2 LDS kernels (A/B reference)
    // freed ciphertext / plaintext arrays are kept per size and handed out again: every op of a context is ordered on its
    // stream, so reuse needs no synchronisation, while hipFree / hipMalloc cost ~60 / ~25 us and a device-wide sync each
    // (a LoLa inference allocates and frees ~300 temporaries per plaintext prime)
    std::unordered_map<size_t, std::vector<uint64_t *>> pool;
    size_t pool_bytes = 0, pool_max;
    void *slabs = nullptr;                // std::vector<Slab>*: small arrays are carved out of slabs (one hipMalloc per <= 64 arrays), cn_api.hip
    bool ks_split14 = true;   // N = 16384: key switch as two 8192-point halves per limb (no register spills); 0 = fused 1024-thread kernel
    bool ks_pair14 = true;    // ... both halves in ONE launch per rotation (k_keyswitch_pair14, round 5); 0 = k_keyswitch_split14 + k_ks_combine14 (+ k_galois_lds)
    bool ks_chain = true;     // SumAllSlots at N = 16384: every link of the rotate-and-add chain hands sigma_next(c1) to the next one (no permutation pass between links)
    int ks_wide = -1;         // -1 auto (small batches), 0 fused kernel, 1 two-launch with a workgroup per digit, 2 two-launch per source limb
    void *ks_part = nullptr; size_t ks_part_cap = 0;   // its partial products [ct][digit][2][k][N]
    char *pin = nullptr, *pin_dev = nullptr; size_t pin_off = 0; uint32_t pin_laps = 0;           // ring of pinned host memory for the small table uploads (cn_api.hip: pin_block)
    char *stage = nullptr; size_t stage_cap = 0;       // staging arena of the deferred per-ciphertext rotations / plaintext products (gather, batched call, scatter)
    int ks_xcd = 0;           // cn_set_option("ks_xcd", v) / CN_KS_XCD=v: fused key switch, workgroup order: 0 (ciphertext, limb); 1 the k workgroups of a
    int gemm_order = 1;                          // scalar GEMM (VALU kernels): 1 = slice-major workgroup order (every input slice fetched once per XCD), 0 = group-major
    bool ks_perm_fused = true;                    // rotations through the two-launch key switch apply the automorphism while loading (no k_galois_lds pass)
    int stream_tries = 0;                        // streams created until one had a hardware queue of its own (cn_api.hip: pick_stream)
    std::atomic<int> probe_pins{0};              // > 0: a cn_ctx_create on this device is measuring this context's stream; cn_ctx_destroy waits
                              // ciphertext on one XCD (share its source limbs in that L2); 2 limb-major (one key slice per XCD L2 at a time)
    bool ks_tight = false;    // CN_KS_TIGHT=1: 128-VGPR key-switch variant (2 workgroups per CU, accumulators spill to scratch)
    std::atomic<bool> capturing{false};   // between cn_graph_begin and cn_graph_end: work is recorded on the stream, nothing that synchronises or allocates may run
    std::vector<std::unique_ptr<char[]>> cap_staged;                 // host blocks of the upload nodes recorded so far
    std::vector<std::pair<uint64_t *, size_t>> cap_allocs;           // arrays handed out while recording
    int graphs_alive = 0;     // graphs carry the addresses of the scratch arenas: those must not move while one exists
    bool mp_fused = true;     // dense MultiplyPlain as k_lift_ntt + k_mul_plain_fused; cn_set_option("mp_fused", 0) = the six separate launches
    bool mp_bcast = true;     // ... one ciphertext x many plaintexts (row-dot batches) as ONE launch that transforms the plaintexts (k_mul_plain_bcast, round 5); 0: the two launches above
    bool gemm_pair = true;    // planned scalar GEMMs: gather lists that share at least half of their inputs are merged in pairs (pair_gather_lists); cn_set_option("gemm_pair", 0): the caller's lists
    bool gemm_mfma = true;    // scalar GEMMs with >= 16 outputs per gather list on the int8 matrix cores (exact); cn_set_option("gemm_mfma", 0): FP64 kernel
    int cus = 0;              // compute units of the device
    bool sq_lds = true;       // fused squaring with the NTT-form operand parked in LDS (N <= 8192) - HBM traffic = the algorithmic 2 reads + 3 writes per
                              // block (profiles/r02_pmc_square_gemm.txt); cn_set_option("sq_lds", 0): parked in the outputs' place (two workgroups per CU)
    uint64_t folded_zero = 0;  // zero encryptions folded so far
    bool fold_zero = true;    // queued fresh encryptions of zero whose only reader is a queued scalar product and which have been released: folded by linearity (k_encrypt_fold, round 6); cn_set_option("fold_zero", 0): materialised
    int enc_fused = 2;        // 2 (default, round 6): a block per (ciphertext, component, limb) - k_encrypt_split, two workgroups per CU: 420 against 542 us per 784 ciphertexts; 1: Encryptor.Encrypt behind the samplers as one kernel (k_encrypt_fused, N <= 8192); cn_set_option("enc_fused", 0): expand + batched transform + k_encrypt_tail
    int sq_pipe = 1;          // 1: fused squaring of a batch (>= 4 blocks per resident workgroup) on the pipelined resident kernel k_square_pipe; 0: k_square_fused; 2: k_square_pipe for any count (tests)
    uint64_t uid = 0;         // creation order within the process (cn_ctx_create)
    bool defer_stagger = false; hipEvent_t ev_front = nullptr;   // deferred flush of a big Multiply + Relinearize group: its Multiply waits for the front of the context that flushed one last
                                                                // on this device (cn_defer.hip: staggered plaintext-prime channels); three forms measured, none with a gain - OFF by default: cn_set_option("defer_stagger", 1) / CN_DEFER_STAGGER=1
    int sq_halves = 1;        // Multiply + Relinearize of >= 512 ciphertexts (N <= 8192) as two halves software-pipelined over the context's two streams: the Multiply of the second half beside the key
                              // switch of the first (pipelined_halves, cn_api_shared.h).  1 (default): the batched entry point cn_mul_relin; 2: also the queued per-ciphertext calls of a flush (measured slower there); 0: off
    bool sq_overlap = false;  // squaring of a batch: the q-side transform kernel on a second stream beside [k_behz_extend -> Bsk side] (cn_eval.hip: do_multiply); CN_SQ_OVERLAP / cn_set_option
    hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; bool stream2_failed = false;
    bool sq_fused = true;     // squarings: forward transforms + tensor + inverse transforms in one kernel; cn_set_option("sq_fused", 0) = separate launches
    // deferred submission (cn_set_option("defer", 1)): per-ciphertext calls are queued and flushed as batched launches; 2: ... and submitted without the
    // context lock through `ring` (cn_submit.h), executed by whoever drains it
    std::atomic<int> defer{0};
    DeferQueue *dq = nullptr;
    SubmitRing *ring = nullptr;
    ReadyRing *ready = nullptr;
    int async_rc = 0; std::string async_msg;        // first error of a record executed from the ring: reported by the next synchronising call
};

// ---------------------------------------------------------------- kernel launchers (cn_l_*.hip)
enum { POL_U64 = 0, POL_F64 = 1, POL_F64L = 2 };       // ArU64 (Shoup, any modulus), ArF64 (< 2^49.4), ArF64L (<= 44 bits: no forward recentring)

// One key switch of `cnt` ciphertexts: out[ct] = (add0[ct], add1[ct]) + KeySwitch(target[ct]) (+ extra[ct] - the fused accumulator of
// the cn_*_add entry points).  target / add0 / add1 / extra are strided per ciphertext (in words); out is dense size-2, or - out_tab -
// one address per ciphertext (deferred per-ciphertext calls whose results live in separate arrays).
// per-ciphertext operands of a two-launch key switch that rotates every ciphertext by its own step count (cn_rotate_rows_many): the ciphertext it reads
// (c0 at in, c1 behind it), the Galois key of its element, the element
struct KsItem { const uint64_t *in; const void *key; uint32_t elt, pad; };
struct KsArgs {
    const uint64_t *target; size_t tstride;
    const uint64_t *add0, *add1; size_t astride;
    const uint64_t *key; uint64_t *out; uint32_t cnt; int galois;
    const uint64_t *extra; size_t xstride;
    uint32_t accmax;          // FP64 accumulators: terms between recentrings
    int mode;                 // 0 fused, 1 two launches / workgroup per digit, 2 two launches / workgroup per source limb
    uint64_t *const *out_tab;
    uint32_t xcd_cts = 0;     // fused kernel: ciphertexts (a multiple of 8) placed XCD-aware, see k_keyswitch_rr
    const KsItem *items = nullptr;   // two-launch variants: per-ciphertext (operand, key, element) table instead of target / add0 / key / perm_elt
    uint32_t perm_elt = 0;    // two-launch variants: Galois element of a rotation whose automorphism the kernels apply while loading (target / add0 = the UNPERMUTED c1 / c0)
                              // k_keyswitch_pair14: target = sigma(c1) already, add0 = the UNPERMUTED c0 (permuted through LDS by the workgroup that owns the limb)
    uint32_t next_elt = 0; uint64_t *next_out = nullptr;   // k_keyswitch_pair14: the new c1 leaves a second time as sigma_next(c1) -> next_out[ct][k][N] (rotate-and-add chains)
};
// k_encrypt_fold (cn_k_rr.hip.h): a scalar-product output that receives the weighted sum of `count` folded zero encryptions, and the terms of all outputs
struct FoldOut { uint64_t *out; uint32_t first, count; };               // terms [first, first + count) of the term table
struct FoldTerm { double w; uint32_t enc, pad; };                       // centred weight (|w| <= t/2), index of the encryption in the samplers' int8 arrays
struct RrOps {                // register-radix kernels of one arithmetic policy; every launcher returns false when the size has no kernel
    int (*set_attrs)(uint32_t logn, size_t lds);
    bool (*ntt)(cn_ctx *c, uint64_t *data, uint32_t limbs, uint32_t base_off, uint32_t nmod, int inverse);
    bool (*intt_tensor)(cn_ctx *c, const uint64_t *A, const uint64_t *B, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm);
    bool (*square_fused)(cn_ctx *c, const uint64_t *A, size_t astride, const uint64_t *const *atab, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm);   // FP64 policies
    bool (*mul_plain_fused)(cn_ctx *c, const uint64_t *pt, uint32_t pitch, uint32_t npt, uint64_t *lift, const uint64_t *src, size_t sstride, uint32_t pstride,
                            uint64_t *out, uint32_t count, uint32_t polys);
    bool (*enc_tail)(cn_ctx *c, const uint64_t *u, const uint64_t *pt, uint32_t pts, uint64_t *out, uint32_t cnt, const int8_t *noise, const void *tab);   // U64, F64; tab: EncTab[cnt] or null
    bool (*enc_fused)(cn_ctx *c, const int8_t *us, const uint64_t *pt, uint32_t pts, uint64_t *out, uint32_t cnt, const int8_t *noise, const void *tab);  // all policies, N <= 8192: u (int8) -> transform -> both components in one kernel
    bool (*mul_plain_bcast)(cn_ctx *c, const uint64_t *pt, uint32_t pitch, const uint64_t *ctn, uint64_t *out, uint32_t count, uint32_t polys, uint32_t next_elt,
                            uint64_t *next_out);                                       // ONE ciphertext (NTT form) x count plaintexts (+ sigma_next(c1) of every product on the side)
    bool (*enc_fold)(cn_ctx *c, const int8_t *us, const int8_t *noise, const void *fout, const void *terms, uint32_t outputs);   // FP64 policies, N <= 8192: k_encrypt_fold
};
struct KsOps {
    int (*set_attrs)(uint32_t logn, size_t lds);
    bool (*launch)(cn_ctx *c, const KsArgs &a);                        // fused / two-phase by a.mode
    bool (*split14)(cn_ctx *c, const KsArgs &a);                       // N = 16384 as two 8192-point halves (FP64 policies), k_ks_combine14 behind it
    bool (*pair14)(cn_ctx *c, const KsArgs &a);                        // N = 16384 in one launch: both halves per (ciphertext, limb) workgroup (FP64 policies)
};
extern const RrOps cn_rr_u64, cn_rr_f64, cn_rr_f64l;
extern const KsOps cn_ks_u64, cn_ks_f64, cn_ks_f64l;

// BEHZ element-wise steps (cn_l_behz.hip); src_tab: one source address per ciphertext instead of src + ct*stride*2kN
int cn_l_behz_extend(cn_ctx *c, const uint64_t *src, uint32_t stride, const uint64_t *const *src_tab, uint64_t *aq, uint64_t *ab, uint32_t cnt);
int cn_l_behz_floor(cn_ctx *c, const uint64_t *dq, const uint64_t *db, uint64_t *out, uint32_t cnt);

// scalar GEMM (cn_l_gemm.hip).  Relative addressing: input / output ciphertext = base + index * ctw; absolute (ABS): the tables hold
// device addresses (deferred per-ciphertext calls: every ciphertext is its own array), 0 = padded tap / no output.
struct GemmLaunch {
    bool small, two, abs; uint32_t MT;
    const uint64_t *in; const void *idx; const void *W; const void *oidx; const uint64_t *bias; const void *bidx; uint64_t *out;
    uint32_t G, M, K, lazy, Kp, obase;
    uint32_t P = 0, mtiles = 0, ksteps = 0;          // matrix-core kernel: weight digit planes, 32-row output tiles, 32-term K steps
    uint32_t polys = 2;                              // ciphertext size of inputs and outputs (3: unrelinearized products)
    uint32_t order = 0;                              // workgroup order of the VALU kernels: 0 group-major, 1 slice-major (gemm_block_coords)
    bool one = false;                                // k_scalar_gemm_f64<MT, 1, 0>: the words are not split (gemm_one_limb)
    uint32_t in_unit = 0, out_unit = 0, bias_unit = 0;   // words per index unit of the table-driven kernels; 0 = one ciphertext / one plaintext (plans).  Deferred calls: 32 (256 B offsets from the lowest address)
};
// k_scalar_gemm_f64: rows of the weight table per (group, output tile) - K terms + zero rows up to a multiple of 16 + 16: the kernel's sets of 4 (or 8) terms run past K
// and multiply whatever word the padded gather entry reads by these zeros
inline uint32_t gemm_f64_rows(uint32_t K) { return ((K + 15) & ~15u) + 16; }
int cn_l_gemm(cn_ctx *c, const GemmLaunch &g);
int cn_l_gemm_mfma(cn_ctx *c, const GemmLaunch &g);   // k_scalar_gemm_mfma: W = weight digit fragments, idx rows of ksteps * 32 entries

inline void cn_launch_count(cn_ctx *c, int n = 1) { c->st.kernel_launches += n; }
