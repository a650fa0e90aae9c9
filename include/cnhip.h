/*
 * cnhip.h -- C ABI of libcnhip.so: the MI355X-native BFV evaluator that replaces the
 * SEAL 3.2 native calls issued by microsoft/CryptoNets' AtomicSealBfvEncryptedVector.
 *
 * Boundary being replaced: the managed->native P/Invoke layer inside SEALNet.dll
 * (NuGet Microsoft.Research.SEALNet 3.2.0, `HE Wrapper/packages.config:6`), reached only
 * from `HE Wrapper/AtomicSealBfvVector.cs` through `epenv.evaluator.*`.  Each entry point
 * below cites the reference call sites (file:line under /root/reference) it serves.
 *
 * Conventions:
 *  - plain C, opaque context pointer + 64-bit buffer handles, no C++/torch types;
 *  - every function returns 0 on success, <0 on error; cn_last_error() returns a
 *    thread-local message (the reference throws .NET exceptions at the same points);
 *  - all functions are thread-safe (one mutex per context: the reference calls from
 *    Defaults.ThreadCount threads, `HE Wrapper/Utils.cs:46-88`);
 *  - work is enqueued on the context's HIP stream; results are ordered; host reads
 *    (cn_ct_download) synchronise; cn_sync() waits for everything;
 *  - BATCHED BY CONSTRUCTION: a buffer handle is an ARRAY of ciphertexts (or dense
 *    plaintexts); every op takes (handle, first index, count), so one layer is a handful
 *    of launches instead of 10^5 P/Invokes.  count==1 is the per-ciphertext SEAL call.
 *  - ciphertext layout = SEAL's: [poly][limb][N] u64 canonical residues, coefficient
 *    form; plaintext = N u64 coefficients mod t; keys = NTT form (bit-reversed order):
 *    key-switch key = for limb l, digit d (low->high): [2][k][N], flattened in (l,d) order.
 */
#ifndef CNHIP_H
#define CNHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cn_ctx cn_ctx;
typedef uint64_t cn_handle;

#define CN_OK 0
#define CN_ERR_ARG (-1)      /* bad argument / shape mismatch (reference: `throw new Exception(...)`) */
#define CN_ERR_HIP (-2)      /* HIP runtime failure */
#define CN_ERR_NOKEY (-3)    /* relin / Galois key missing (SEAL: "Galois key not present") */
#define CN_ERR_ZERO (-4)     /* multiply_plain by an all-zero plaintext (SEAL: "plain cannot be zero") */
#define CN_ERR_NODEV (-5)    /* no HIP device available */

int cn_version(void);
const char *cn_last_error(void);
int cn_device_count(void);

/* ---- context = AtomicSealBfvEncryptedEnvironment (AtomicSealBfvVector.cs:19-74,140-173):
 * (n, coeff moduli q[k], plain modulus t, DecompositionBitCount, GaloisDecompositionBitCount). */
int cn_ctx_create(uint32_t n, const uint64_t *q, uint32_t k, uint64_t t, int dbc, int gdbc,
                  int device, cn_ctx **out);
int cn_ctx_destroy(cn_ctx *ctx);
int cn_sync(cn_ctx *ctx);
/* Ordering between contexts without a host wait: everything submitted to `ctx` AFTER this call starts only when everything submitted to
 * `other` BEFORE it has finished (an event on other's stream that ctx's stream waits for; same or different device).  The plaintext-prime
 * channels of one vector are independent contexts (EncryptedSealBfvVector.cs:225-236); a host that issues them from one thread uses this to
 * stagger them - the FP64-bound key switch of one channel then runs beside the HBM-bound layers of the other instead of beside its twin
 * (bench.py --stagger).  Neither context is synchronised with the host. */
int cn_ctx_wait_for(cn_ctx *ctx, cn_ctx *other);
/* tuning switches (A/B testing): "f64" = 1 (default) runs transforms of moduli < 2^49 and key switching in exact FP64
 * (set BEFORE uploading keys), 0 = integer Shoup path everywhere; "legacy_ntt" = 1 selects the radix-2 LDS kernels;
 * "ks_wide" = -1 (default: automatic by batch size) / 0 fused one-launch kernel / 1 two launches with one workgroup per digit
 * (1-2 ciphertexts; 1-6 until round 3) / 2 two launches with one workgroup per source limb (3-32 ciphertexts);
 * "ks_tight" = 1 the 128-VGPR fused variant; "ks_split14" = 1 (default) runs the N = 16384 key switch as two 8192-point
 * halves per limb (no register spills), 0 = the fused 1024-thread kernel; "sq_fused" = 1 (default) runs the transforms and the tensor
 * of a squaring (Multiply(a, a): SquareActivation) as one kernel per base, 0 = separate launches; "mp_fused" = 1 (default) runs a dense
 * MultiplyPlain as two launches (lift + transform of the plaintexts; transform, product, inverse transform of the ciphertext limbs),
 * 0 = six; "sq_lds" = 1 (default) parks the NTT-form operand of a fused squaring in LDS (N <= 8192), 0 = in the outputs' place; "sq_pipe" = 1 (default) runs the fused squaring of a batch on the
 * pipelined resident kernel (k_square_pipe: one workgroup per CU and modulus, inverse root table in LDS, next operand prefetched), 0 = k_square_fused; "enc_fused" = 2 (default, round 6) runs Encryptor.Encrypt behind the samplers as ONE kernel with a block per (ciphertext, component, limb) (N <= 8192,
 * FP64 policies: two workgroups per CU), 1 = a block per (ciphertext, limb) (the ternary u goes from the sampler's int8 polynomial through one transform and stays in registers
 * for both components), 0 = expansion + batched transform + tail kernel;
 * "gemm_mfma" = 1 (default) runs wide scalar GEMMs (cn_scalar_gemm / cn_scalar_dot batches with >= 16 outputs) on the int8 matrix
 * cores; "gemm_pair" = 1 (default) lets cn_scalar_gemm / cn_gemm_plan_create merge gather lists that share at least half of their inputs in pairs
 * (small signed weights, lists of <= 64 entries and <= 5 outputs: the windows of a convolution - every shared input then travels to a CU once, not twice),
 * 0 = the caller's lists as they are (environment: CN_GEMM_PAIR; CN_GEMM_ONE_LIMB=0 keeps the small-weight kernel on its two-limb form).
 * All variants produce identical words. */
/* "defer" = 1: DEFERRED SUBMISSION for callers that issue one evaluator call per ciphertext from many threads - the unchanged
 * NeuralNetworks layers of the reference (PoolLayer.cs:67-80,113-121,182,214; EncryptedSealBfvMatrix.cs:79-120,140-154; LLInterleaveLayer.cs;
 * Utils.cs:46-88).  cn_scalar_dot, cn_add, cn_sub, cn_add_plain, cn_mul_relin, cn_encrypt and - on up to 4 ciphertexts per call - cn_mul_plain,
 * cn_rotate_rows(_add), cn_rotate_columns(_add), cn_sum_slots, cn_copy are then queued with their operand addresses, ordered by data dependence,
 * and launched as batched kernels (all pending calls of one dependency level, kind and parameter = one launch chain) at layer boundaries (a
 * scalar product or multiplication that reads the queued result of another one), when the queue is full, or when any other entry point
 * (cn_sync, downloads, ...) needs the results.  Same words as immediate calls; argument errors (ranges, zero plaintexts, missing Galois keys) are
 * reported by the call that made them, device errors by the call that triggered the flush.  cn_free of a handle with pending readers is safe
 * (the array returns to the pool after the flush).
 * "defer" = 2 (round 6): the same queue, fed WITHOUT THE CONTEXT LOCK.  cn_scalar_dot, cn_add, cn_sub, cn_add_plain, cn_mul_relin (up to 4 ciphertexts per
 * call), cn_encrypt (up to 4), cn_encrypt_zero_new, cn_free, cn_free_many and cn_ct_alloc(1, 2) do not take the lock: a call claims the next slot of a
 * multi-producer ring with one atomic add, writes a 64-byte record (handles and indices as passed) and returns 0; whoever finds the lock free executes
 * the published records in claim order - a total order consistent with happens-before between the caller's threads, which is what the dependence
 * tracking needs.  Single-ciphertext allocations come from a ring of ready handles the executing thread keeps filled.  Every other entry point takes the
 * lock and first executes what was published before it.  DIFFERENCE TO "defer" = 1: the arguments of a published call are checked when it is executed -
 * an error is reported ONCE by the next call that synchronises with the context (cn_sync, downloads, cn_stats_get, any non-deferrable entry point:
 * "a call submitted without the lock (defer = 2) failed ..."), the calls around it are executed.  For callers that come from Defaults.ThreadCount =
 * Environment.ProcessorCount threads (HE Wrapper/Defaults.cs:11-15, Utils.cs:46-88): the unchanged CryptoNets layers run at the same rate from 4, 16 and
 * 256 threads.  cn_live_handles does not count the ready handles; "ready_handles" reads their number; "pin_laps" the laps of the context's pinned
 * upload ring (small tables of a flush; 32 MiB, CN_PIN_RING_MIB in the environment; a lap waits for the stream once).
 * "fold_zero" = 1 (default, round 6): a queued fresh encryption of zero (cn_encrypt with pt = 0 / cn_encrypt_zero_new) that only feeds ONE queued scalar
 * product and has been released by the caller (PoolLayer.ElementAt / ReleaseTemp, PoolLayer.cs:67-90) is not materialised: sum_t w_t Enc_t(0) is
 * added onto the scalar product's output by linearity - exact modular arithmetic on the same sampler draws (nonce, item), the SAME words as with
 * "fold_zero" = 0, a fifth of the transforms.  All or nothing per flush (every queued zero encryption must qualify).  "folded_zero_encryptions" reads the count.
 * "defer_stagger" = 0 (default; 1: the Multiply of a queued squaring group of >= 256 ciphertexts waits on the device for the Multiply of an older context of the same device -
 * measured without a gain, profiles/r06_stagger_ab.txt).
 * "sq_halves" = 1 (default, round 6; 0 off; 2: also inside the flush of queued per-ciphertext calls - measured slower there): cn_mul_relin of >= 512 ciphertexts at
 * N <= 8192 runs as two halves software-pipelined over two streams of the context - the Multiply of the second half beside the key switch of the first (its HBM-bound base
 * extension / floor fill what the FP64-bound key switch leaves).  Same words; every later call on the context is ordered behind both halves.  A caller that issues its plaintext
 * primes one after the other (or from parallel tasks) gets what bench.py's half-batch stagger of the primes gets (12.6 -> 12.0 ms per CryptoNets batch); a staggered caller nothing.
 * "sq_overlap" = 0 (default; 1: the q-side transform kernel of a batched squaring on a second stream beside the base extension - measured slower in
 * the two-context batch, profiles/r06_square_overlap.txt).
 * "gemm_order" = 1 (default): slice-major workgroup order of the VALU scalar GEMM (every input slice fetched once per XCD), 0 = group-major.
 * "ks_perm_fused" = 1 (default): a rotation of a small batch (two-launch key switch) has no permutation pass - the key-switch kernels apply the
 * automorphism while they load c1 and c0; 0 = k_galois_lds in front of them.  "stream_tries" (read only): streams cn_ctx_create tried until one had a
 * hardware queue of its own (< 0: none had; CN_STREAM_PROBE=0 takes the first).
 * Environment switches read once per process (A/B measurements, all default to the measured best): CN_STREAM_PROBE=0 (no hardware-queue selection),
 * CN_TABLES_ZERO_COPY=0 (small operand tables are copied to the device instead of read from the pinned ring), CN_KS_WIDE_MAX / CN_KS_DIGIT_MAX ((ciphertext,
 * limb) blocks up to which a key switch runs as two launches: 160 / with one workgroup per digit: 10), CN_GEMM_ORDER, CN_DEFER_TRACE=1 (one stderr line
 * per flushed queue level: calls per kind, launches).
 * "ks_xcd": workgroup order of the fused key switch - 0 (ciphertext, limb), 1 the limbs of a ciphertext on one XCD (default at N = 16384), 2 limb-major
 * (default up to N = 8192: one key slice per XCD L2 at a time).
 * N = 16384, batches (round 5): "ks_pair14" = 1 (default) runs a key switch as ONE launch - both 8192-point halves of a limb in one workgroup, the last
 * inverse stage and the addends applied on the way out, a rotation's c1 permuted once per ciphertext and its c0 inside the kernel; 0 = the three launches of
 * rounds 1-4 (permutation pass, two workgroups per limb, combining pass).  "ks_chain" = 1 (default): every link of a cn_sum_slots / cn_rowdot_batch
 * rotate-and-add chain leaves the NEXT link's permuted c1 beside its result (no permutation pass between links).  "mp_bcast" = 1 (default): cn_rowdot_batch
 * transforms its ONE ciphertext once and the row plaintexts inside the product kernel (one launch); 0 = k_lift_ntt + k_mul_plain_fused.  Environment:
 * CN_KS_PAIR14, CN_KS_CHAIN (the same switches for a whole process), CN_DEFER_MERGE_GEMM=0 (deferred scalar products of one flush keep their levels:
 * one launch per level instead of one per term count).  Environment: CN_LOCK_GRACE_NS / CN_LOCK_COMBINE switch the two context-lock experiments that
 * are kept but off (cn_host.cpp).
 * "ks_xi": DECOMPOSITION CONVENTION of the key switch (relinearisation and rotations).  0 (default): base-2^dbc digits of the raw residue c_l of every
 * source limb l; key (l, d) = (-(a s + e) + 2^(dbc d) s' [in limb l only], a) - SURVEY 9.5, the form in which the CRT basis element
 * (q/q_l) [(q/q_l)^-1]_{q_l} is folded into the key.  1: digits of xi_l = [c_l (q/q_l)^-1]_{q_l}; key (l, d) = (-(a s + e) + (q/q_l) 2^(dbc d) s' [in every
 * limb], a) - the xi_q decomposition as the BEHZ paper writes it.  Both are exact key switches and decrypt identically with their own keys; keys of
 * one convention give garbage under the other.  Set it BEFORE cn_keygen / the key uploads; the start-up self-test of the host mirrors
 * (hewrapper.AtomicSealBfvEncryptedEnvironment.SelfTest, the C# twin's SelfTest) picks the one the client's evaluator obeys. */
int cn_set_option(cn_ctx *ctx, const char *name, int value);
/* reads a switch back, or a choice the library made: "behz_small_base" (1: auxiliary primes below 2^49 - the FP64 kernels - k+1 of them,
 * or k+2 where k+1 are too few (N = 16384); 0: SEAL's 61-bit base, taken whenever log2 t + log2 N + log2 q + 2 < log2(B m_sk) does not
 * hold for the small primes or a data prime has 49 bits or more), "behz_f64", "aux_primes" (primes of B plus m_sk), "pending_calls" (deferred calls not yet launched), "f64", "defer", "ks_wide", "ks_xi", "ks_pair14", "ks_chain", "mp_bcast", "sq_fused", "sq_pipe", "enc_fused", "mp_fused" */
int cn_get_option(cn_ctx *ctx, const char *name, int *value);
/* SEAL DefaultParams.CoeffModulus128(n) (AtomicSealBfvVector.cs:146); returns count, fills q (<=9) */
int cn_default_coeff_modulus(uint32_t n, uint64_t *q);
/* number of u64 words of one key-switch key for this context (relin: which=0, galois: which=1) */
size_t cn_key_words(cn_ctx *ctx, int which);
/* keys.RelinKeys(dbc) / keys.GaloisKeys(gdbc) (AtomicSealBfvVector.cs:68-69): upload from host
 * memory, or adopt a device buffer (e.g. one filled by an RCCL broadcast). */
int cn_set_relin_key(cn_ctx *ctx, const uint64_t *words, size_t count, int is_device_ptr);
int cn_set_galois_key(cn_ctx *ctx, uint64_t galois_elt, const uint64_t *words, size_t count, int is_device_ptr);
int cn_has_galois_key(cn_ctx *ctx, uint64_t galois_elt);
/* Any key in either representation.  which: 0 relin, 1 galois (galois_elt), 2 public, 3 secret - as cn_get_key; a public / secret key is always copied.
 * form 0: NTT form in this library's transform order (what cn_set_relin_key / cn_set_galois_key / cn_set_public_key take: SEAL's in-memory form IF its
 * transform uses the minimal primitive 2N-th root and bit-reversed output).  form 1: COEFFICIENT form - the device transforms the polynomials
 * with its own tables, so the upload does not depend on the root or the output order of the key generator's transform (an adopted device buffer is
 * transformed in place).  The C# twin's start-up self-test falls back to form 1 (Evaluator.TransformFromNTTInplace on the key ciphertexts) when the
 * NTT-form words of the SEAL it runs beside do not reproduce SEAL's own results (integration/GpuAtomicSealBfvEncryptedVector.cs: SelfTest). */
int cn_load_key(cn_ctx *ctx, int which, uint64_t galois_elt, const uint64_t *words, size_t count, int is_device_ptr, int form);
/* Multi-GPU, single process (the path shards by independent batches / plaintext primes, the only exchange is the one-time
 * key broadcast): copies the relinearisation key and every Galois key of ctxs[0] into ctxs[1..n-1] (same encryption parameters, any
 * devices) - ONE RCCL broadcast per key over xGMI to the contexts on other GPUs (librccl is loaded on demand; peer copies without it),
 * device-to-device copies on the root's GPU.  Synchronises all contexts.  (One process per GPU: broadcast with the host framework and
 * adopt the buffers through cn_set_relin_key / cn_set_galois_key with is_device_ptr = 1 instead - bench.py does.) */
int cn_ctx_broadcast_keys(cn_ctx **ctxs, int n);
/* Evaluator/util galois_elt_from_step: steps>0 left, <0 right, 0 = column swap (2N-1) */
uint64_t cn_galois_elt_from_step(cn_ctx *ctx, int steps);

/* ---- buffers = arrays of SEAL Ciphertext / Plaintext objects (ctor/Set/Dispose:
 * AtomicSealBfvVector.cs:373,389-397,413-428) */
int cn_ct_alloc(cn_ctx *ctx, uint32_t count, uint32_t size, cn_handle *out);   /* size = polys per ct (2 or 3) */
int cn_pt_alloc(cn_ctx *ctx, uint32_t count, cn_handle *out);                  /* dense plaintexts, N coeffs each */
int cn_free(cn_ctx *ctx, cn_handle h);
/* n handles with one call (one lock acquisition): ReleaseTemp of the unchanged PoolLayer disposes one zero encryption per padded tap (PoolLayer.cs:83-90),
 * BaseLayer.GetNext one column at a time (BaseLayer.cs:23-49).  All handles are validated first; an invalid or repeated one releases nothing. */
int cn_free_many(cn_ctx *ctx, const cn_handle *h, uint32_t n);
int cn_ct_upload(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, const uint64_t *host);
int cn_ct_download(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, uint64_t *host);
int cn_pt_upload(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, const uint64_t *host);
int cn_pt_download(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, uint64_t *host);
/* BatchEncoder.Encode / Decode (AtomicSealBfvVector.cs:644,669,941,1130,1158 / :1050,1093): slot values <-> plaintext
 * coefficients; the (I)NTT mod t runs on the device.  Requires t prime, t == 1 mod 2N. */
int cn_encode(cn_ctx *ctx, const uint64_t *values, uint32_t nvalues, cn_handle pt, uint32_t pi);
int cn_decode(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint64_t *values /* N */);
/* the same for `count` plaintexts in one call (one upload, one scatter launch, one batched transform mod t): values is
 * [count][nvalues] for encode (slots beyond nvalues are zero), [count][N] for decode.  A layer's weight rows / masks / bias vectors
 * (LLDenseLayer.Prepare: thousands of Plaintexts, EncryptedSealBfvMatrix.cs:79-120) are encoded with ONE call. */
int cn_encode_batch(cn_ctx *ctx, const uint64_t *values, uint32_t nvalues, uint32_t count, cn_handle pt, uint32_t pi);
int cn_decode_batch(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint32_t count, uint64_t *values /* [count][N] */);
int cn_copy(cn_ctx *ctx, cn_handle src, uint32_t sfirst, cn_handle dst, uint32_t dfirst, uint32_t count);
/* n single ciphertexts (or dense plaintexts) that live in n arrays -> consecutive places of ONE array with one launch:
 * dst[dfirst + i] = src[i][sfirst[i]] (sfirst NULL = element 0 of every array).  The reference keeps a vector as a list of Ciphertext objects and
 * has no such call; a host that keeps vectors as arrays uses it where the reference copies element by element (GenerateSparseOfArray,
 * AtomicSealBfvVector.cs:1347-1359; the column gather in front of DenseMatrixBySparseVectorMultiply): 25 + 10 copy launches of one LoLa image
 * become 2.  Queued like n cn_copy calls under cn_set_option("defer", 1). */
int cn_copy_many(cn_ctx *ctx, const cn_handle *src, const uint32_t *sfirst, uint32_t n, cn_handle dst, uint32_t dfirst);
int cn_device_ptr(cn_ctx *ctx, cn_handle h, void **ptr, size_t *bytes);
int cn_live_handles(cn_ctx *ctx);                                              /* leak counter */

/* ---- linear ops ---------------------------------------------------------------------- */
/* Evaluator.Add / Sub / Negate (AtomicSealBfvVector.cs:491,646,671,865,1005,1258) */
int cn_add(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count);
int cn_sub(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count);
int cn_negate(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle out, uint32_t oi, uint32_t count);
/* Evaluator.AddMany (AtomicSealBfvVector.cs:502,698,708,902): out[oi] = sum_i in[idx[i]] */
int cn_add_many(cn_ctx *ctx, cn_handle in, const uint32_t *idx, uint32_t n_idx, cn_handle out, uint32_t oi);
/* Evaluator.AddPlain / SubPlain (AtomicSealBfvVector.cs:1019,1267), dense plaintexts */
int cn_add_plain(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract,
                 cn_handle out, uint32_t oi, uint32_t count);
/* Evaluator.MultiplyPlain with a dense (BatchEncoded) plaintext: lift, NTT, dyadic, INTT
 * (AtomicSealBfvVector.cs:645,670,803,855,942,1455).  pt_stride 0 = same plaintext for all. */
int cn_mul_plain(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, uint32_t pt_stride,
                 cn_handle out, uint32_t oi, uint32_t count);
/* Evaluator.MultiplyPlain with a constant (sparse-format) plaintext `Plaintext(hex)`
 * (AtomicSealBfvVector.cs:1136,1164,1179 -> :472,482,571,592): scalars[i] in [0,t). */
int cn_mul_scalar(cn_ctx *ctx, cn_handle a, uint32_t ai, const uint64_t *scalars, uint32_t scalar_stride,
                  cn_handle out, uint32_t oi, uint32_t count);
/* HOT LOOP A: AtomicSealBfvEncryptedVector.DenseMatrixBySparseVectorMultiply for O outputs at
 * once (AtomicSealBfvVector.cs:434-521 as driven by PoolLayer.ConvolveOnce/Apply,
 * NeuralNetworks/PoolLayer.cs:113-121,149-229):
 *   out[oi+o] = sum_k W[o*K+k] * in[idx[o*K+k]]  (+ Delta-scaled dense plaintext bias[bias_idx[o]])
 * idx<0 = padded tap (skipped, PoolLayer.cs:68-80); W in [0,t), zero weights skipped (:468);
 * an output whose weights are all zero is an error like SEAL's AddMany of nothing.
 * `in` and `out` must have the same ciphertext size: 2, or 3 (Evaluator.MultiplyPlain / Add on products that have not been relinearized;
 * cn_gemm_plan_apply likewise - a plan does not depend on the size). */
int cn_scalar_gemm(cn_ctx *ctx, cn_handle in, const int32_t *idx, const uint64_t *W, uint32_t O, uint32_t K,
                   cn_handle bias_pt, const int32_t *bias_idx, cn_handle out, uint32_t oi);
/* The same product for ONE output whose K input ciphertexts are SEPARATE objects - the form the C# twin of
 * AtomicSealBfvEncryptedVector.DenseMatrixBySparseVectorMultiply calls per block i (AtomicSealBfvVector.cs:434-521: denses[k].encData[i]
 * are individually allocated SEAL Ciphertexts): out[oi] = sum_k w[k] * in[k][in_idx[k]].  in[k] == 0: padded tap, skipped
 * (PoolLayer.cs:68-80); in_idx == NULL: all 0; zero weights contribute nothing (:468); an all-zero row is an error.  With
 * cn_set_option("defer", 1) the call is queued (see cn_set_option) and merged with its siblings into one GEMM launch. */
int cn_scalar_dot(cn_ctx *ctx, const cn_handle *in, const uint32_t *in_idx, const uint64_t *w, uint32_t K, cn_handle out, uint32_t oi);
/* The same GEMM planned once: validation, grouping and the weight tiles in kernel layout are built and uploaded by
 * cn_gemm_plan_create (the layer's weights then live in HBM), cn_gemm_plan_apply only launches.  Release with cn_free. */
int cn_gemm_plan_create(cn_ctx *ctx, const int32_t *idx, const uint64_t *W, uint32_t O, uint32_t K, cn_handle bias_pt,
                        const int32_t *bias_idx, cn_handle *plan);
int cn_gemm_plan_apply(cn_ctx *ctx, cn_handle plan, cn_handle in, cn_handle out, uint32_t oi);

/* Captured sequences (HIP graphs) for the launch-bound chains of a single-image inference (LoLa: ~235 small launches per plaintext
 * prime, the reference's LowLatencyCryptoNets loop LoLaCryptonets.cs:236-278): every library call between cn_graph_begin and
 * cn_graph_end is RECORDED on the context stream instead of executed; cn_graph_launch replays the whole sequence with one launch.
 * Rules: run the same sequence once before recording (the scratch arenas then have their size; temporaries come out of the handle pool);
 * nothing that synchronises (cn_sync, uploads / downloads, key changes) between begin and end; handles created while recording stay
 * alive as long as the graph is launched (its kernels carry their addresses); new inputs are written INTO the handles the recorded
 * sequence read (cn_copy, cn_encrypt).  Release with cn_free. */
int cn_graph_begin(cn_ctx *ctx);
int cn_graph_end(cn_ctx *ctx, cn_handle *graph);
int cn_graph_launch(cn_ctx *ctx, cn_handle graph);

/* ---- non-linear ops ------------------------------------------------------------------ */
/* Evaluator.Multiply (BEHZ), size2 x size2 -> size3 (AtomicSealBfvVector.cs:461,546,786,839,1457) */
int cn_multiply(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out3, uint32_t oi, uint32_t count);
/* Evaluator.Relinearize size3 -> size2 (AtomicSealBfvVector.cs:462,547,787,840) */
int cn_relinearize(cn_ctx *ctx, cn_handle in3, uint32_t ii, cn_handle out, uint32_t oi, uint32_t count);
/* HOT LOOP B: Multiply + Relinearize per block (PointwiseMultiply, AtomicSealBfvVector.cs:839-840;
 * SquareActivation.cs:10-13).  a_stride/b_stride 0 broadcast one operand (PointwiseMultiplySparseDimOne). */
int cn_mul_relin(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t a_stride, cn_handle b, uint32_t bi, uint32_t b_stride,
                 cn_handle out, uint32_t oi, uint32_t count);

/* ---- rotations (HOT LOOP C) ----------------------------------------------------------- */
/* Evaluator.ApplyGalois: automorphism + key switch of c1 */
int cn_apply_galois(cn_ctx *ctx, cn_handle in, uint32_t ii, uint64_t galois_elt, cn_handle out, uint32_t oi, uint32_t count);
/* Evaluator.RotateRows(/Inplace): NAF decomposition when no key exists for the step
 * (AtomicSealBfvVector.cs:625,631,637,660,864,1420,1458).  Operand and result ranges of ONE handle may be the same range (in place), disjoint, or
 * overlap with a shift (cn_rotate_rows, cn_apply_galois, cn_rotate_columns: such a call takes the permutation pass / a staging copy, which reads the whole
 * operand before anything is written).  The _add forms refuse a partial overlap of operand or accumulator with the result (CN_ERR_ARG): their fused
 * accumulator is read where the result is stored. */
int cn_rotate_rows(cn_ctx *ctx, cn_handle in, uint32_t ii, int steps, cn_handle out, uint32_t oi, uint32_t count);
/* RotateRows of n ciphertexts by n DIFFERENT step counts: out[oi[i]] = RotateRows(in[ii[i]], steps[i]), same words as n cn_rotate_rows calls.
 * The reference rotates the vectors of an Interleave / a Vectorize one Evaluator.RotateRows at a time (AtomicSealBfvVector.cs:628-688); here the
 * hops of all n rotations run in rounds - one two-launch key switch per round over every ciphertext that has a hop left, each with the key
 * and the element of ITS step count - so a single-image network pays the dispatches of the longest rotation instead of the sum (n <= 32 / k; more
 * run one after the other).  in == out is allowed when no result overwrites another rotation's operand or result (its own is fine). */
int cn_rotate_rows_many(cn_ctx *ctx, cn_handle in, const uint32_t *ii, const int *steps, uint32_t n, cn_handle out, const uint32_t *oi);
/* Evaluator.RotateColumns(/Inplace) (AtomicSealBfvVector.cs:709,914,1391) */
int cn_rotate_columns(cn_ctx *ctx, cn_handle in, uint32_t ii, cn_handle out, uint32_t oi, uint32_t count);
/* out[i] = acc[i] + RotateRows(in[i], steps) / + RotateColumns(in[i]): the rotate-and-add step of SumAllSlots / RotateRowsAndAdd
 * (AtomicSealBfvVector.cs:862-868, 888-955) with the addition fused into the last key-switch kernel; acc and in may alias
 * out.  Same words as the rotation followed by cn_add. */
int cn_rotate_rows_add(cn_ctx *ctx, cn_handle in, uint32_t ii, int steps, cn_handle acc, uint32_t ai, cn_handle out, uint32_t oi, uint32_t count);
int cn_rotate_columns_add(cn_ctx *ctx, cn_handle in, uint32_t ii, cn_handle acc, uint32_t ai, cn_handle out, uint32_t oi, uint32_t count);
/* HOT LOOP C in one call.  cn_sum_slots: SumAllSlots(length) (AtomicSealBfvVector.cs:888-935) of `count` single-block ciphertexts
 * h[first..], in place: column swap + add when length >= N/2, then RotateRows(-2^s) + AddInplace for 2^s < length; length 0 = all
 * slots.  cn_rowdot_batch: out[oi + r] = SumAllSlots(v[vi] * pt[pi + r], length), r < rows - every row of a row-major plaintext
 * matrix against one packed ciphertext (EncryptedSealBfvMatrix.cs:79-120, LLDenseLayer / LLPackedDenseLayer /
 * LLInterleavedDenseLayer).  Same words as the per-row MultiplyPlain / RotateRows / Add sequence of the reference. */
int cn_sum_slots(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, uint32_t length);
int cn_rowdot_batch(cn_ctx *ctx, cn_handle v, uint32_t vi, cn_handle pt, uint32_t pi, uint32_t rows, uint32_t length, cn_handle out, uint32_t oi);

/* ---- client side on the device (what SEAL's KeyGenerator / Encryptor / Decryptor do for
 * AtomicSealBfvEncryptedEnvironment.SetKeys / Encrypt / Decrypt, AtomicSealBfvVector.cs:62-74,1030-1110,1202-1232), for data
 * owners that have a GPU.  Randomness: a ChaCha20 counter-mode generator (256-bit key per context, 64-bit nonce per call). ---- */
/* Sampler: ChaCha20 (RFC 7539 block function) in counter mode.  cn_set_rng_key installs the 256-bit key of the context (the data owner draws
 * it from the OS entropy source; default all zero = reproducible, for tests); the `seed` argument of cn_keygen / cn_encrypt is the 64-bit
 * nonce of the call - distinct calls under one key need distinct nonces (a counter or fresh entropy).  cn_set_rng_salt sets only the first
 * 64 key bits (kept for callers of the round-1 interface). */
int cn_set_rng_key(cn_ctx *ctx, const uint8_t *key32);
int cn_set_rng_salt(cn_ctx *ctx, uint64_t salt);
/* known-answer self-test of the generator: one raw block (16 words) for a key, the 64-bit block counter (state words 12-13) and the
 * 64-bit nonce (state words 14-15); RFC 7539 section 2.3.2 is the case counter = 0x0900000000000001, nonce = 0x4a000000 */
int cn_rng_selftest(cn_ctx *ctx, const uint8_t *key32, uint64_t counter, uint64_t nonce, uint32_t *out16);
int cn_keygen(cn_ctx *ctx, uint64_t seed, int with_galois);          /* secret, public, relin (dbc) and default Galois (gdbc) keys */
int cn_set_public_key(cn_ctx *ctx, const uint64_t *words, size_t count);   /* [2][k][N], NTT form */
int cn_set_secret_key(cn_ctx *ctx, const uint64_t *words, size_t count);   /* [k][N], NTT form */
int cn_get_key(cn_ctx *ctx, int which /*0 relin,1 galois,2 public,3 secret*/, uint64_t galois_elt, uint64_t *host, size_t count);
/* Encryptor.Encrypt of `count` dense plaintexts (pt = 0: encryptions of zero; pt_stride 0: the same plaintext).  With
 * cn_set_option("defer", 1) a call for up to 4 ciphertexts is queued like the evaluator calls (the unchanged PoolLayer encrypts a zero
 * vector per padded convolution tap, PoolLayer.cs:67-80: 645 calls per layer and plaintext prime become one launch chain). */
int cn_encrypt(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint32_t pt_stride, cn_handle out, uint32_t oi, uint32_t count, uint64_t seed);
/* AllocateCiphertext + Encryptor.Encrypt(PlainZero) as ONE call: a new one-ciphertext array holding a fresh encryption of zero (PoolLayer.ElementAt per
 * padded tap, PoolLayer.cs:67-80; the IsZero branches of AtomicSealBfvVector.cs:566,587).  Same words and - under "defer" - the same queue entry as
 * cn_ct_alloc followed by cn_encrypt(pt = 0, count = 1, seed). */
int cn_encrypt_zero_new(cn_ctx *ctx, uint64_t seed, cn_handle *out);
/* Decryptor.Decrypt of size-2 or size-3 ciphertexts into dense plaintexts */
int cn_decrypt(cn_ctx *ctx, cn_handle ct, uint32_t ci, uint32_t count, cn_handle pt_out, uint32_t pi);
/* Decryptor.InvariantNoiseBudget as CryptoTracker.TestBudget probes it (HE Wrapper/CryptoTracker.cs:41-52, BaseLayer.cs:37): writes the
 * residues of t*(c0 + c1 s + c2 s^2) mod q_j, [count][k][N], to `host`; the caller composes the limbs (CRT) and takes
 * budget = log2(q) - log2(centred infinity norm) - 1.  Needs the secret key (client-side context); synchronises. */
int cn_noise_poly(cn_ctx *ctx, cn_handle ct, uint32_t ci, uint32_t count, uint64_t *host);

/* ---- raw transforms (kernel benchmarks / parity tests of the NTT itself) --------------- */
/* in-place negacyclic NTT over `limbs` limbs of N words at a device pointer; limb i uses modulus
 * (i % nmod) of base 0 (coeff moduli q) or base 1 (BEHZ Bsk moduli).  Async on the ctx stream. */
int cn_ntt_forward(cn_ctx *ctx, void *dev_ptr, uint32_t limbs, int base);
int cn_ntt_inverse(cn_ctx *ctx, void *dev_ptr, uint32_t limbs, int base);
int cn_ct_ntt(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, int inverse);   /* same on a ct array */
/* times `iters` back-to-back launches of the forward NTT kernel with HIP events on the ctx
 * stream; returns average milliseconds per launch in *ms. */
int cn_ntt_time(cn_ctx *ctx, void *dev_ptr, uint32_t limbs, int base, int inverse, int iters, float *ms);
/* VALU issue rate of the device at this moment: ns per wave-instruction per SIMD, measured with `launches` launches of a kernel that only
 * runs arithmetic chains (512-thread workgroups, one per CU, two waves per SIMD - the fused key switch's occupancy).  kind 0: FP64, chains
 * of the exact modular multiply (`iters` x 48 instructions per thread); kind 1: full-rate 32-bit integer VALU (`iters` x 24).  bench.py
 * prices the key switch's instruction counts with them. */
int cn_valu_issue_time(cn_ctx *ctx, int kind, int iters, int launches, float *ns_per_instr);
void *cn_stream(cn_ctx *ctx);                       /* hipStream_t of the context */
int cn_event_time_begin(cn_ctx *ctx);               /* HIP-event stopwatch on the ctx stream */
int cn_event_time_end(cn_ctx *ctx, float *ms);

/* ---- statistics = OperationsCount (AtomicSealBfvVector.cs:211-294) ---------------------- */
typedef struct cn_stats {
    uint64_t Multiplication, PlainMultiplication, Addition, PlainAddition, Subtraction, PlainSubtraction,
        Rotation, AddMany, AddManyItemCount, Relinarization;   /* reference names */
    uint64_t ntt_forward_limbs, ntt_inverse_limbs, kernel_launches;
} cn_stats;
int cn_stats_get(cn_ctx *ctx, cn_stats *out, int reset);

#ifdef __cplusplus
}
#endif
#endif
