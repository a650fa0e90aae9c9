// GPU twin of `HE Wrapper/AtomicSealBfvVector.cs` (microsoft/CryptoNets): the SAME classes - AtomicSealBfvEncryptedEnvironment,
// AtomicSealBfvEncryptedVector, OperationsCount - with the SAME members, so that EncryptedSealBfvVector.cs, EncryptedSealBfvMatrix.cs,
// IFactory.cs, CryptoTracker.cs, Utils.cs and everything in NeuralNetworks/ compile against it UNCHANGED.  Build `HE Wrapper` with the
// symbol CNHIP defined, this file + CnHip.cs + GpuSealBfvFactory.cs added and AtomicSealBfvVector.cs wrapped in `#if !CNHIP`
// (INTEGRATION.md has the .csproj lines).  Every `epenv.evaluator.*` call of the reference (SURVEY.md section 2a lists them with
// their lines) becomes a libcnhip call on device-resident ciphertexts; what SEAL does on the CLIENT side stays SEAL: KeyGenerator,
// Encryptor, Decryptor, BatchEncoder (AtomicSealBfvVector.cs:62-74, 1042, 1130, 1211) - the evaluation device never sees a secret key.
//
// Not compiled in this repository (no .NET toolchain in the build image); tests/test_integration_cs.py checks that every P/Invoke it
// uses exists in include/cnhip.h with the generated signature and that every member the unchanged reference files use is defined here.
//
// Storage: a vector's blocks live in ONE device array (`enc`: handle + count); plaintext vectors keep their SEAL Plaintexts on the host
// (they are created by SEAL's BatchEncoder) and a lazily uploaded device copy for AddPlain / MultiplyPlain.  Calls arrive from
// Defaults.ThreadCount threads (Utils.ParallelProcessInEnv): libcnhip serialises per context and - with deferred submission, on by
// default here - merges the per-ciphertext calls of a layer into a few batched launches (include/cnhip.h, cn_set_option "defer").
#if CNHIP
using System;
using System.Collections.Concurrent;
using System.Collections.Generic;
using System.Diagnostics;
using System.IO;
using System.Linq;
using System.Numerics;
using System.Reflection;
using System.Threading;
using System.Threading.Tasks;
using MathNet.Numerics.LinearAlgebra;
using Microsoft.Research.SEAL;

namespace HEWrapper
{
    /// <summary>One libcnhip context (= one plaintext modulus on one GPU), shared by every environment copy like the reference shares
    /// its Evaluator and keys (AtomicSealBfvVector.cs:45-60).  Destroyed by the finalizer of the last reference.</summary>
    public sealed class CnDevice
    {
        public IntPtr Ctx { get; private set; }
        public readonly uint N, K;
        public readonly int DeviceIndex;
        public CnDevice(EncryptionParameters parms, int dbc, int gdbc, int deviceIndex, bool deferred)
        {
            var q = parms.CoeffModulus.Select(m => m.Value).ToArray();
            N = (uint)parms.PolyModulusDegree; K = (uint)q.Length; DeviceIndex = deviceIndex;
            CnHip.Check(CnHip.cn_ctx_create(N, q, K, parms.PlainModulus.Value, dbc, gdbc, deviceIndex, out IntPtr c));
            Ctx = c;
            // 2: the deferrable calls (the per-ciphertext calls of the unchanged layers) are PUBLISHED to the context's submission ring without taking its lock and
            // executed in claim order by whoever finds the lock free (libcnhip round 6: 0.87-0.90 of the batched rate at 16 AND at 256 caller threads, 0.69-0.78 under
            // the lock).  Their argument errors surface at the next synchronising call (Download / Decrypt / cn_sync) - the twin checks its arguments itself.
            CnHip.Check(CnHip.cn_set_option(Ctx, "defer", deferred ? 2 : 0));
            LockFree = deferred;
        }
        /// <summary>deferred calls go through the lock-free submission ring: a release is ONE published record - nothing is parked (the library then sees every
        /// Dispose where the caller made it and knows which zero vectors of PoolLayer.ElementAt are dead when the layer is flushed: it folds them)</summary>
        public readonly bool LockFree;
        ~CnDevice() { if (Ctx != IntPtr.Zero) { CnHip.cn_ctx_destroy(Ctx); Ctx = IntPtr.Zero; } }
        public int CtWords(int size = 2) { return (int)(size * K * N); }

        // Disposal in batches.  The unchanged layers dispose one vector at a time from many threads (PoolLayer.ReleaseTemp: one Dispose per zero
        // encryption, PoolLayer.cs:83-90; BaseLayer.GetNext: the columns of a layer's input, BaseLayer.cs:23-49): a released handle is parked in one of 64
        // lists OF THIS DEVICE (picked by the calling thread's id: threads rarely meet on a list) and a list goes to the library 32 handles at a time
        // (cn_free_many: one lock acquisition instead of 32).  What is left in the lists only holds device memory - at most 64 x 31 handles; it is released
        // when results leave the device (Download / Decrypt call FlushFrees) and dies with the device: nothing static refers to a CnDevice (ADVICE r04: the
        // round-4 thread-static dictionaries in a static bag kept every device - its HBM, pool and keys - alive for the life of the process).
        const int FreeBatch = 32, Stripes = 64;
        readonly List<ulong>[] parked = Enumerable.Range(0, Stripes).Select(_ => new List<ulong>(FreeBatch)).ToArray();
        public void DeferFree(ulong handle)
        {
            if (LockFree) { if (Ctx != IntPtr.Zero) CnHip.cn_free(Ctx, handle); return; }
            var list = parked[Thread.CurrentThread.ManagedThreadId & (Stripes - 1)];
            ulong[] batch = null;
            lock (list)
            {
                list.Add(handle);
                if (list.Count >= FreeBatch) { batch = list.ToArray(); list.Clear(); }
            }
            if (batch != null && Ctx != IntPtr.Zero) CnHip.cn_free_many(Ctx, batch, (uint)batch.Length);
        }
        /// <summary>releases the parked handles of every thread for this device</summary>
        public void FlushFrees()
        {
            foreach (var list in parked)
            {
                ulong[] batch;
                lock (list)
                {
                    if (list.Count == 0) continue;
                    batch = list.ToArray();
                    list.Clear();
                }
                if (Ctx != IntPtr.Zero) CnHip.cn_free_many(Ctx, batch, (uint)batch.Length);
            }
        }
    }

    /// <summary>A device array of ciphertexts (libcnhip handle).  Released by Dispose or by the finalizer (AtomicSealBfvVector.cs:379-404).</summary>
    public sealed class CnBuffer : IDisposable
    {
        public readonly CnDevice Dev; public ulong Handle; public readonly uint Count;
        public CnBuffer(CnDevice dev, uint count, bool plain = false)
        {
            Dev = dev; Count = count;
            if (plain) CnHip.Check(CnHip.cn_pt_alloc(dev.Ctx, count, out Handle)); else CnHip.Check(CnHip.cn_ct_alloc(dev.Ctx, count, 2, out Handle));
        }
        ~CnBuffer() { Free(); }
        /// <summary>adopts an array the library allocated itself (cn_encrypt_zero_new)</summary>
        public CnBuffer(CnDevice dev, ulong handle, uint count) { Dev = dev; Handle = handle; Count = count; }
        /// <summary>release at once instead of parking the handle: set on the result of DenseMatrixBySparseVectorMultiply - the deferred queue folds the
        /// `conv.Add(bias)` that follows (PoolLayer.cs:184-186) into the scalar product only for an intermediate it KNOWS to be dead</summary>
        public bool FreeNow;
        void Free()
        {
            if (Handle == 0 || Dev.Ctx == IntPtr.Zero) return;
            if (FreeNow) CnHip.cn_free(Dev.Ctx, Handle); else Dev.DeferFree(Handle);
            Handle = 0;
        }
        public void Dispose() { Free(); GC.SuppressFinalize(this); }
    }

    /// <summary>SEAL 3.2 objects <-> u64 words.  A SEAL Ciphertext is [poly][limb][N] u64 words (the layout of include/cnhip.h).  Words are
    /// read and written through the PUBLIC indexer of Ciphertext (`this[ulong]` get / set, UInt64Count, Resize) - the members SEALNet's own
    /// Save / Load are written with; every SEALNet member this file touches is listed, with its source, in integration/SEALNET_MANIFEST.json
    /// (checked by tests/test_integration_cs.py).  (Round 2 read the words out of the Save() stream behind an assumed 73-byte header: no
    /// faster - SEALNet's Save walks the same indexer - and not backed by anything in this repository; removed.)</summary>
    public static class SealInterop
    {
        public static ulong[] Words(Ciphertext c)
        {
            ulong count = c.UInt64Count;
            var w = new ulong[count];
            for (ulong i = 0; i < count; i++) w[i] = c[i];
            return w;
        }
        public static Ciphertext ToCiphertext(ulong[] words, AtomicSealBfvEncryptedEnvironment env, bool nttForm = false)
        {
            ulong n = env.parameters.PolyModulusDegree, k = (ulong)env.parameters.CoeffModulus.Count();
            ulong size = (ulong)words.Length / (n * k);
            var c = new Ciphertext(env.context, env.memoryPool);
            c.Resize(env.context, env.context.FirstParmsId, size);
            c.IsNTTForm = nttForm;
            for (ulong i = 0; i < (ulong)words.Length; i++) c[i] = words[i];
            return c;
        }
        /// <summary>N coefficients of a BatchEncoded plaintext (shorter plaintexts are zero padded)</summary>
        public static ulong[] Coeffs(Plaintext p, uint n)
        {
            var w = new ulong[n];
            ulong cnt = Math.Min(p.CoeffCount, (ulong)n);
            for (ulong i = 0; i < cnt; i++) w[i] = p[i];
            return w;
        }
        /// <summary>the constant of a sparse-format plaintext `Plaintext(hex)` (AtomicSealBfvVector.cs:1136)</summary>
        public static ulong Constant(Plaintext p) { return p.CoeffCount == 0 ? 0UL : p[0]; }
        /// <summary>Key-switch key for libcnhip: SEAL 3.2 keeps one Ciphertext (size 2, NTT form, [2][k][N] words) per (source limb l, digit d),
        /// in the order relinearize_one_step / apply_galois walk them - l outer, d inner, digits low to high (SEAL 3.2 evaluator.cpp).  That
        /// is the [(l,d)][2][k][N] layout of cn_set_relin_key / cn_set_galois_key: the words are concatenated as they are.</summary>
        public static ulong[] KeySwitchKey(IEnumerable<Ciphertext> digits)
        {
            var parts = digits.Select(Words).ToList();
            var all = new ulong[parts.Sum(p => p.Length)];
            int pos = 0;
            foreach (var p in parts) { Array.Copy(p, 0, all, pos, p.Length); pos += p.Length; }
            return all;
        }
    }

    public class AtomicSealBfvEncryptedEnvironment : IComputationEnvironment
    {
        // --- client side: SEAL, exactly as in the reference (AtomicSealBfvVector.cs:22-37)
        public Evaluator evaluator;               // kept for source compatibility; the hot path never touches it
        public Encryptor encryptor;
        public Decryptor decryptor;
        public EncryptionParameters parameters = null;
        public SEALContext context = null;
        public readonly MemoryPoolHandle memoryPool = MemoryPoolHandle.New();
        public SecretKey secretKey;
        public PublicKey publicKey;
        public RelinKeys relinKeys;
        public BatchEncoder builder;
        public GaloisKeys galoisKeys;
        public Plaintext PlainZero;
        public ulong plainmodulusValue = 0;
        public int PlaintextCapacity { get { return (int)parameters.PolyModulusDegree; } }
        public ulong CiphertextCapacity { get { return 3; } }
        public IFactory ParentFactory { get; set; }
        // --- evaluation side: the device context (shared by all copies of this environment)
        public CnDevice device;
        /// <summary>GPU the NEXT environment is created on / whether its calls are queued and batched (GpuSealBfvFactory sets both)</summary>
        public static int DefaultDeviceIndex = 0;
        public static bool DeferredSubmission = true;
        int decompositionBitCount, galoisDecompositionBitCount;
        long encryptNonce;                        // nonce of the next device encryption (any value, never reused under one sampler key)
        public ulong NextNonce() { return (ulong)Interlocked.Increment(ref encryptNonce); }

        public AtomicSealBfvEncryptedEnvironment() { }
        public AtomicSealBfvEncryptedEnvironment(AtomicSealBfvEncryptedEnvironment p)
        {
            parameters = p.parameters; context = p.context; builder = p.builder; relinKeys = p.relinKeys; secretKey = p.secretKey;
            publicKey = p.publicKey; evaluator = p.evaluator; encryptor = p.encryptor; decryptor = p.decryptor; galoisKeys = p.galoisKeys;
            PlainZero = p.PlainZero; ParentFactory = p.ParentFactory; plainmodulusValue = p.plainmodulusValue;
            device = p.device;                    // (device encryptions of a copy draw their nonces from the parent: see EncryptZeroInto)
            nonceSource = p.nonceSource ?? p;
        }
        AtomicSealBfvEncryptedEnvironment nonceSource = null;
        public ulong DeviceNonce() { return (nonceSource ?? this).NextNonce(); }

        /// <summary>uploads the PUBLIC evaluation keys into HBM (cn_set_relin_key / cn_set_galois_key); the secret key stays on the host</summary>
        void CreateDevice(int dbc, int gdbc)
        {
            decompositionBitCount = dbc; galoisDecompositionBitCount = gdbc;
            device = new CnDevice(parameters, dbc, gdbc, DefaultDeviceIndex, DeferredSubmission);
            // the PUBLIC key goes to the device too: zero vectors (PoolLayer.ElementAt encrypts one per padded convolution tap, PoolLayer.cs:67-80;
            // the IsZero branches of the multiply-by-plain methods) are encrypted there - cn_encrypt(pt = 0), queued and batched like the
            // evaluator calls - instead of SEAL Encrypt on the host + a 640 KiB upload each.  PublicKey.Data is a size-2 NTT-form Ciphertext:
            // [2][k][N] words, the layout of cn_set_public_key.  The sampler's 256-bit key comes from the OS.
            var pkw = SealInterop.Words(publicKey.Data);
            CnHip.Check(CnHip.cn_set_public_key(device.Ctx, pkw, (UIntPtr)pkw.Length));
            var rngKey = new byte[32];
            using (var osRng = System.Security.Cryptography.RandomNumberGenerator.Create()) osRng.GetBytes(rngKey);
            CnHip.Check(CnHip.cn_set_rng_key(device.Ctx, rngKey));
            // first nonce: 8 bytes of their OWN from the OS generator - a nonce is public material (it travels in logs and traces), so it must not
            // be derived from the sampler key
            var nonceBytes = new byte[8];
            using (var osRng = System.Security.Cryptography.RandomNumberGenerator.Create()) osRng.GetBytes(nonceBytes);
            encryptNonce = BitConverter.ToInt64(nonceBytes, 0);
            // RelinKeys.Data[0]: the key that takes a size-3 ciphertext to size 2 (relinearize_one_step)
            var rk = SealInterop.KeySwitchKey(relinKeys.Data.First());
            CnHip.Check(CnHip.cn_set_relin_key(device.Ctx, rk, (UIntPtr)rk.Length, 0));
            // GaloisKeys.Data[(elt - 1) / 2]: the key of Galois element elt (empty when the key was not generated)
            ulong index = 0;
            foreach (var key in galoisKeys.Data)
            {
                var digits = key.ToList();
                if (digits.Count > 0)
                {
                    var gk = SealInterop.KeySwitchKey(digits);
                    CnHip.Check(CnHip.cn_set_galois_key(device.Ctx, 2 * index + 1, gk, (UIntPtr)gk.Length, 0));
                }
                index++;
            }
            SelfTest();
        }

        /// <summary>what the last SelfTest settled on: "ks_xi=0|1 keys=ntt|coeff" (null: not run)</summary>
        public string SelfTestReport { get; internal set; }
        /// <summary>set to false to skip the start-up self-test (a few SEAL evaluator calls + device calls per environment, ~0.1 s)</summary>
        public static bool RunSelfTest = true;

        /// <summary>Start-up agreement check of the device with the SEAL it is dropped into.  The environment keeps a live SEAL Evaluator
        /// (AtomicSealBfvVector.cs:22,64): two fresh ciphertexts go through it and through libcnhip - MultiplyPlain (dense, constant), AddPlain,
        /// Multiply, then the key-switching operations Relinearize, RotateRows(1), RotateRows(-1), RotateColumns - and the ciphertext WORDS are
        /// compared.  SEAL 3.2 is an un-vendored dependency of the reference, so two properties of its keys could not be read off a source file:
        /// the decomposition convention of the key switch (cn_set_option "ks_xi": digits of the raw residue with the message term in limb l only,
        /// or digits of [c_l (q/q_l)^-1]_{q_l} with the RNS image of (q/q_l) 2^(dbc d) s' - non-zero in limb l only - in the key) and the order of the NTT-form key words.  On a key-switch
        /// mismatch the test flips "ks_xi", then re-uploads the keys in COEFFICIENT form (Evaluator.TransformFromNTTInplace on copies of the key
        /// ciphertexts; cn_load_key form 1 - the device transforms them with its own tables) and tries both conventions again.  The first
        /// combination that reproduces SEAL's words is kept; if none does, or if an operation without keys disagrees in its DECRYPTED SLOTS (words that differ
        /// while the slots agree are a warning in SelfTestReport: another valid representative, not an interoperability property), it throws and names the
        /// operation - a wrong recollection of SEAL becomes an exception at start-up, not rc 0 and garbage in the middle of an inference.
        /// Python mirror with the same procedure: cryptonets_amd/hewrapper.py AtomicSealBfvEncryptedEnvironment.SelfTest (tests/test_self_test.py).</summary>
        public void SelfTest()
        {
            if (!RunSelfTest || evaluator == null || encryptor == null) return;
            uint n = device.N;
            ulong t = plainmodulusValue;
            var v0 = Enumerable.Range(0, (int)n).Select(i => ((ulong)i * 2654435761UL + 12345UL) % t).ToList();
            var v1 = Enumerable.Range(0, (int)n).Select(i => ((ulong)i * 40503UL + 7UL) % t).ToList();
            using (var p0 = new Plaintext(memoryPool)) using (var p1 = new Plaintext(memoryPool))
            using (var ca = new Ciphertext(context, memoryPool)) using (var cb = new Ciphertext(context, memoryPool))
            using (var c3 = new Ciphertext(context, memoryPool)) using (var want = new Ciphertext(context, memoryPool))
            {
                builder.Encode(v0, p0); builder.Encode(v1, p1);
                encryptor.Encrypt(p0, ca, memoryPool); encryptor.Encrypt(p1, cb, memoryPool);
                ulong constant = p1[1] == 0 ? 1UL : p1[1];
                var pc = new Plaintext(constant.ToString("X"), memoryPool);
                ulong h2, h3, o2, pt;
                CnHip.Check(CnHip.cn_ct_alloc(device.Ctx, 2, 2, out h2)); CnHip.Check(CnHip.cn_ct_alloc(device.Ctx, 1, 3, out h3));
                CnHip.Check(CnHip.cn_ct_alloc(device.Ctx, 1, 2, out o2)); CnHip.Check(CnHip.cn_pt_alloc(device.Ctx, 1, out pt));
                try
                {
                    CnHip.Check(CnHip.cn_ct_upload(device.Ctx, h2, 0, 1, SealInterop.Words(ca)));
                    CnHip.Check(CnHip.cn_ct_upload(device.Ctx, h2, 1, 1, SealInterop.Words(cb)));
                    CnHip.Check(CnHip.cn_pt_upload(device.Ctx, pt, 0, 1, SealInterop.Coeffs(p1, n)));
                    Func<ulong, int, ulong[]> read = (h, size) =>
                    {
                        var w = new ulong[device.CtWords(size)];
                        CnHip.Check(CnHip.cn_ct_download(device.Ctx, h, 0, 1, w));
                        return w;
                    };
                    Func<string, ulong[]> dev = op =>
                    {
                        switch (op)
                        {
                            case "MultiplyPlain": CnHip.Check(CnHip.cn_mul_plain(device.Ctx, h2, 0, pt, 0, 1, o2, 0, 1)); break;
                            case "MultiplyPlain(constant)": CnHip.Check(CnHip.cn_mul_scalar(device.Ctx, h2, 0, new ulong[] { constant }, 0, o2, 0, 1)); break;
                            case "AddPlain": CnHip.Check(CnHip.cn_add_plain(device.Ctx, h2, 0, pt, 0, 0, o2, 0, 1)); break;
                            case "Multiply": CnHip.Check(CnHip.cn_multiply(device.Ctx, h2, 0, h2, 1, h3, 0, 1)); return read(h3, 3);
                            case "Relinearize":
                                CnHip.Check(CnHip.cn_multiply(device.Ctx, h2, 0, h2, 1, h3, 0, 1));
                                CnHip.Check(CnHip.cn_relinearize(device.Ctx, h3, 0, o2, 0, 1)); break;
                            case "RotateRows(1)": CnHip.Check(CnHip.cn_rotate_rows(device.Ctx, h2, 0, 1, o2, 0, 1)); break;
                            case "RotateRows(-1)": CnHip.Check(CnHip.cn_rotate_rows(device.Ctx, h2, 0, -1, o2, 0, 1)); break;
                            case "RotateColumns": CnHip.Check(CnHip.cn_rotate_columns(device.Ctx, h2, 0, o2, 0, 1)); break;
                        }
                        return read(o2, 2);
                    };
                    Func<string, ulong[]> seal = op =>
                    {
                        switch (op)
                        {
                            case "MultiplyPlain": evaluator.MultiplyPlain(ca, p1, want, memoryPool); break;
                            case "MultiplyPlain(constant)": evaluator.MultiplyPlain(ca, pc, want, memoryPool); break;
                            case "AddPlain": evaluator.AddPlain(ca, p1, want); break;
                            case "Multiply": evaluator.Multiply(ca, cb, want, memoryPool); break;
                            case "Relinearize": evaluator.Multiply(ca, cb, c3, memoryPool); evaluator.Relinearize(c3, relinKeys, want, memoryPool); break;
                            case "RotateRows(1)": evaluator.RotateRows(ca, 1, galoisKeys, want, memoryPool); break;
                            case "RotateRows(-1)": evaluator.RotateRows(ca, -1, galoisKeys, want, memoryPool); break;
                            case "RotateColumns": evaluator.RotateColumns(ca, galoisKeys, want, memoryPool); break;
                        }
                        return SealInterop.Words(want);
                    };
                    // Operations without keys: equal WORDS is what the restatement of SEAL 3.2 predicts.  Words that differ while the client's Decryptor returns the
                    // same slots for both are another valid representative (another BEHZ auxiliary base, another lift) - a valid drop-in: a warning in
                    // SelfTestReport, the start-up goes on.  Different slots: fatal.  (Python mirror: hewrapper.py SelfTest; tests/test_self_test.py)
                    Func<ulong[], int, List<ulong>> slots = (words, size) =>       // (size is implied by the word count)
                    {
                        using (var c = SealInterop.ToCiphertext(words, this)) using (var p = new Plaintext(memoryPool))
                        {
                            decryptor.Decrypt(c, p);
                            var vals = new List<ulong>();
                            builder.Decode(p, vals);
                            return vals;
                        }
                    };
                    var warnings = new List<string>();
                    foreach (var op in new[] { "MultiplyPlain", "MultiplyPlain(constant)", "AddPlain", "Multiply" })
                    {
                        var got = dev(op); var exp = seal(op);
                        if (got.SequenceEqual(exp)) continue;
                        int size = op == "Multiply" ? 3 : 2;
                        if (decryptor == null || !slots(got, size).SequenceEqual(slots(exp, size)))
                            throw new Exception(String.Format("libcnhip self-test: {0} differs from SEAL's Evaluator (plaintext modulus {1}), words AND decrypted slots - the device does not implement this SEAL's arithmetic; no key convention can repair that", op, t));
                        warnings.Add(op + ": words differ, decrypted slots equal");
                        Console.Error.WriteLine("libcnhip self-test warning: " + warnings[warnings.Count - 1]);      // never silent (ADVICE r05)
                    }
                    bool rotations = galoisKeys != null && galoisKeys.Data.Any(k => k.Any());
                    var ksOps = rotations ? new[] { "Relinearize", "RotateRows(1)", "RotateRows(-1)", "RotateColumns" } : new[] { "Relinearize" };
                    var wanted = ksOps.ToDictionary(op => op, op => seal(op));
                    // (Relinearize inherits the product's words: compared by slots when the product was only slot-equal)
                    bool relinBySlots = warnings.Any(w => w.StartsWith("Multiply:"));
                    Func<string> firstFailure = () => ksOps.FirstOrDefault(op => op == "Relinearize" && relinBySlots ? !slots(dev(op), 2).SequenceEqual(slots(wanted[op], 2))
                                                                                                                        : !dev(op).SequenceEqual(wanted[op]));
                    var tried = new List<string>();
                    int xi0;
                    CnHip.Check(CnHip.cn_get_option(device.Ctx, "ks_xi", out xi0));
                    foreach (var form in new[] { "ntt", "coeff" })
                    {
                        if (form == "coeff") UploadKeysInCoefficientForm();
                        foreach (int xi in new[] { xi0, 1 - xi0 })
                        {
                            CnHip.Check(CnHip.cn_set_option(device.Ctx, "ks_xi", xi));
                            string bad = firstFailure();
                            tried.Add(String.Format("(ks_xi={0}, keys={1}: {2})", xi, form, bad ?? "ok"));
                            if (bad == null) { SelfTestReport = String.Format("ks_xi={0} keys={1}{2}", xi, form, warnings.Count == 0 ? "" : " warnings: " + String.Join("; ", warnings)); return; }
                        }
                    }
                    CnHip.Check(CnHip.cn_set_option(device.Ctx, "ks_xi", xi0));
                    throw new Exception(String.Format("libcnhip self-test: no key-switch convention reproduces SEAL's Evaluator (plaintext modulus {0}); tried {1}", t, String.Join(" ", tried)));
                }
                finally
                {
                    CnHip.cn_free(device.Ctx, h2); CnHip.cn_free(device.Ctx, h3); CnHip.cn_free(device.Ctx, o2); CnHip.cn_free(device.Ctx, pt);
                    pc.Dispose();
                }
            }
        }

        /// <summary>every evaluation key again, with each key polynomial taken out of SEAL's NTT form by SEAL itself (Evaluator.TransformFromNTTInplace on a
        /// COPY of the key ciphertext) - cn_load_key form 1 then transforms with the device's own tables, whatever root and order SEAL's transform uses</summary>
        void UploadKeysInCoefficientForm()
        {
            Func<IEnumerable<Ciphertext>, ulong[]> coeffWords = digits => SealInterop.KeySwitchKey(digits.Select(c =>
            {
                var copy = new Ciphertext(c);
                evaluator.TransformFromNTTInplace(copy);
                return copy;
            }).ToList());
            var rk = coeffWords(relinKeys.Data.First());
            CnHip.Check(CnHip.cn_load_key(device.Ctx, 0, 0, rk, (UIntPtr)rk.Length, 0, 1));
            ulong index = 0;
            foreach (var key in galoisKeys.Data)
            {
                var digits = key.ToList();
                if (digits.Count > 0)
                {
                    var gk = coeffWords(digits);
                    CnHip.Check(CnHip.cn_load_key(device.Ctx, 1, 2 * index + 1, gk, (UIntPtr)gk.Length, 0, 1));
                }
                index++;
            }
            using (var pkc = new Ciphertext(publicKey.Data))
            {
                evaluator.TransformFromNTTInplace(pkc);
                var pkw = SealInterop.Words(pkc);
                CnHip.Check(CnHip.cn_load_key(device.Ctx, 2, 0, pkw, (UIntPtr)pkw.Length, 0, 1));
            }
        }

        public void SetKeys(KeyGenerator keys, int DecompositionBitCount, int GaloisDecompositionBitCount)
        {
            evaluator = new Evaluator(context);
            encryptor = new Encryptor(context, keys.PublicKey);
            decryptor = new Decryptor(context, keys.SecretKey);
            builder = new BatchEncoder(context);
            relinKeys = keys.RelinKeys(DecompositionBitCount);
            galoisKeys = keys.GaloisKeys(GaloisDecompositionBitCount);
            secretKey = new SecretKey(keys.SecretKey);
            publicKey = new PublicKey(keys.PublicKey);
            PlainZero = new Plaintext("0", memoryPool);
            plainmodulusValue = parameters.PlainModulus.Value;
            CreateDevice(DecompositionBitCount, GaloisDecompositionBitCount);
        }

        public AtomicSealBfvEncryptedEnvironment GetPublicKeys()
        {
            return new AtomicSealBfvEncryptedEnvironment(this) { secretKey = null, decryptor = null };
        }

        public void SaveToFile(string fileName)
        {
            Console.WriteLine("Opening file {0} for writing", fileName);
            using (var file = File.Create(fileName)) SaveToStream(file);
        }

        public void SaveToStream(Stream stream, bool withPrivateKeys = true)
        {
            EncryptionParameters.Save(parameters, stream);
            publicKey.Save(stream);
            relinKeys.Save(stream);
            galoisKeys.Save(stream);
            if (withPrivateKeys) secretKey.Save(stream); else new SecretKey().Save(stream);
        }

        public void LoadFromStream(Stream stream)
        {
            parameters = EncryptionParameters.Load(stream);
            context = SEALContext.Create(parameters);
            publicKey = new PublicKey(); publicKey.Load(context, stream);
            relinKeys = new RelinKeys(); relinKeys.Load(context, stream);
            galoisKeys = new GaloisKeys(); galoisKeys.Load(context, stream);
            secretKey = new SecretKey(); secretKey.Load(context, stream);
            plainmodulusValue = parameters.PlainModulus.Value;
            evaluator = new Evaluator(context);
            encryptor = new Encryptor(context, publicKey);
            try { decryptor = new Decryptor(context, secretKey); }
            catch (Exception)
            {
                decryptor = null;
                Console.WriteLine("WARNING: no Secret Key in file. Will not be able to decrypt messages.");
            }
            builder = new BatchEncoder(context);
            PlainZero = new Plaintext("0", memoryPool);
            CreateDevice(relinKeys.DecompositionBitCount, galoisKeys.DecompositionBitCount);
        }

        public void LoadFromFile(String fileName) { using (var file = File.OpenRead(fileName)) LoadFromStream(file); }

        public static EncryptionParameters Parms(ulong t, ulong n, int SmallModulusCount)
        {
            var parms = new EncryptionParameters(SchemeType.BFV)
            {
                PlainModulus = new SmallModulus(t), PolyModulusDegree = n, CoeffModulus = DefaultParams.CoeffModulus128(n)
            };
            if (SmallModulusCount > 0) parms.CoeffModulus = parms.CoeffModulus.Take(SmallModulusCount).ToList();
            return parms;
        }
        public static EncryptionParameters Parms(ulong t, ulong n, List<SmallModulus> coefModulus = null)
        {
            return new EncryptionParameters(SchemeType.BFV) { PlainModulus = new SmallModulus(t), PolyModulusDegree = n, CoeffModulus = coefModulus };
        }

        public void GenerateEncryptionKeys(ulong prime, ulong n, int DecompositionBitCount, int GaloisDecompositionBitCount, int SmallModulusCount)
        {
            GenerateEncryptionKeys(Parms(prime, n, SmallModulusCount), DecompositionBitCount, GaloisDecompositionBitCount);
        }
        public void GenerateEncryptionKeys(EncryptionParameters parms, int DecompositionBitCount, int GaloisDecompositionBitCount)
        {
            this.parameters = parms;
            context = SEALContext.Create(parameters);
            SetKeys(new KeyGenerator(context), DecompositionBitCount, GaloisDecompositionBitCount);
        }
        public void GenerateEncryptionKeys(string hexPrime, ulong n, int DecompositionBitCount, int GaloisDecompositionBitCount, int SmallModulusCount = -1)
        {
            GenerateEncryptionKeys(Convert.ToUInt64(hexPrime, 16), n, DecompositionBitCount, GaloisDecompositionBitCount, SmallModulusCount);
        }

        ConcurrentQueue<AtomicSealBfvEncryptedEnvironment> environmentQueue = new ConcurrentQueue<AtomicSealBfvEncryptedEnvironment>();
        public IComputationEnvironment AllocateComputationEnv()
        {
            environmentQueue.TryDequeue(out AtomicSealBfvEncryptedEnvironment env);
            return env ?? new AtomicSealBfvEncryptedEnvironment(this);
        }
        public void FreeComputationEnv(IComputationEnvironment env) { environmentQueue.Enqueue(env as AtomicSealBfvEncryptedEnvironment); }
        public void Save(string FileName, bool withPrivateKeys) { throw new NotImplementedException(); }
        public Stream Save(Stream stream, bool withPrivateKeys) { throw new NotImplementedException(); }
        public ulong[] Primes { get { return new ulong[] { plainmodulusValue }; } }
    }

    /// <summary>Operation counters (AtomicSealBfvVector.cs:211-294): the managed counters the layers print in DEBUG builds; the device
    /// keeps the same counters per context (cn_stats_get), CnHip.cn_stats_get(dev.Ctx, out stats, reset) reads them.</summary>
    public static class OperationsCount
    {
        public static int Destructor, Encryption, Plain, Decryption, Multiplication, PlainMultiplication, Addition, Dispose;
        public static int PlainAddition, Subtraction, PlainSubtraction, Rotation, AddMany, AddManyItemCount, Relinarization;
        static Dictionary<string, int> Totals = null;
        static OperationsCount() { AppDomain.CurrentDomain.ProcessExit += (s, e) => PrintTotals(); }
        static IEnumerable<FieldInfo> Counters() { return typeof(OperationsCount).GetFields().Where(f => f.FieldType == typeof(int)); }
        [Conditional("DEBUG")] public static void Add(ref int counter, int value) { Interlocked.Add(ref counter, value); }
        [Conditional("DEBUG")]
        public static void Print()
        {
            if (Totals == null) return;
            Console.WriteLine("Operations:");
            foreach (var f in Counters()) Console.WriteLine("\t{0}\t{1}", f.Name, (int)f.GetValue(null));
        }
        [Conditional("DEBUG")]
        public static void Reset()
        {
            if (Totals == null) Totals = new Dictionary<string, int>();
            foreach (var f in Counters())
            {
                int v = (int)f.GetValue(null);
                Totals[f.Name] = Totals.ContainsKey(f.Name) ? Totals[f.Name] + v : v;
                f.SetValue(null, 0);
            }
        }
        [Conditional("DEBUG")]
        public static void PrintTotals()
        {
            if (Totals == null) return;
            Console.WriteLine("Operations (total):");
            foreach (var f in Counters()) Console.WriteLine("\t{0}\t{1}", f.Name, (int)f.GetValue(null) + Totals[f.Name]);
        }
    }

    /// <summary>A vector under ONE plaintext modulus (AtomicSealBfvVector.cs:303-1475) with its ciphertexts in HBM.</summary>
    public class AtomicSealBfvEncryptedVector : IVector
    {
        CnBuffer enc = null;              // encrypted: one device array with every block (the reference's Ciphertext[] encData)
        Plaintext[] plainData = null;     // plain: SEAL plaintexts (BatchEncoded blocks, or hex constants in sparse format)
        CnBuffer plainDev = null;         // lazily uploaded polynomial form of plainData (AddPlain / MultiplyPlain operands)

        public bool IsSigned { get; set; } = false;
        public EVectorFormat Format { get; set; } = EVectorFormat.dense;
        /// <summary>CryptoTracker.TestVectorBudget (CryptoTracker.cs:78-84) reads `Data as Ciphertext[]`: materialised from the device on demand</summary>
        public object Data { get { return (enc == null) ? (object)plainData : Download(); } }
        internal void RegisterDim(ulong dim) { this.Dim = dim; }
        public ulong Dim { get; private set; } = 0;
        public ulong BlockSize { get { return (enc != null) ? enc.Dev.N : plainData[0].CoeffCount; } }
        public double Scale { get; private set; }
        public bool IsEncrypted { get { return (plainData == null & enc != null); } }
        AtomicSealBfvEncryptedEnvironment owner;      // for Data / Write: the SEAL context the words belong to

        public AtomicSealBfvEncryptedVector(Vector<double> v, IComputationEnvironment env, double Scale = 1.0, bool SignedNumbers = true, bool EncryptData = true, EVectorFormat Format = EVectorFormat.dense)
        {
            this.Scale = Scale; this.IsSigned = SignedNumbers;
            if (EncryptData) Encrypt(v, Format, env); else Plain(v, Format, env);
        }
        public AtomicSealBfvEncryptedVector(UInt64[] v, IComputationEnvironment env, double Scale = 1.0, bool SignedNumbers = true, bool EncryptData = true, EVectorFormat Format = EVectorFormat.dense)
        {
            this.Scale = Scale; this.IsSigned = SignedNumbers;
            if (EncryptData) Encrypt(v, Format, env); else Plain(v, Format, env);
        }
        /// <summary>copy constructor (deep copy, :365-374): cn_copy on the device</summary>
        public AtomicSealBfvEncryptedVector(IVector v, AtomicSealBfvEncryptedEnvironment env)
        {
            var ev = v as AtomicSealBfvEncryptedVector;
            Scale = ev.Scale; Dim = ev.Dim; IsSigned = ev.IsSigned; Format = ev.Format; owner = env;
            if (ev.enc != null)
            {
                enc = new CnBuffer(env.device, ev.enc.Count);
                CnHip.Check(CnHip.cn_copy(env.device.Ctx, ev.enc.Handle, 0, enc.Handle, 0, enc.Count));
            }
            plainData = ev.plainData?.Select(x => { var p = new Plaintext(env.memoryPool); p.Set(x); return p; }).ToArray();
        }
        private AtomicSealBfvEncryptedVector() { }
        static AtomicSealBfvEncryptedVector Result(AtomicSealBfvEncryptedEnvironment env, uint blocks)
        {
            return new AtomicSealBfvEncryptedVector() { owner = env, enc = new CnBuffer(env.device, blocks), plainData = null };
        }

        ~AtomicSealBfvEncryptedVector() { OperationsCount.Add(ref OperationsCount.Destructor, 1); FreeResources(); }
        void FreeResources()
        {
            enc?.Dispose(); plainDev?.Dispose();
            if (plainData != null) foreach (var p in plainData) p.Dispose();
            enc = null; plainDev = null; plainData = null;
        }
        public void Dispose() { FreeResources(); GC.SuppressFinalize(this); OperationsCount.Add(ref OperationsCount.Dispose, 1); }

        // ---------------------------------------------------------------------------------------------------- host <-> device
        Ciphertext[] Download()
        {
            var dev = enc.Dev;
            var words = new ulong[enc.Count * dev.CtWords()];
            CnHip.Check(CnHip.cn_ct_download(dev.Ctx, enc.Handle, 0, enc.Count, words));       // drains the deferred queue, synchronises
            dev.FlushFrees();                                                                  // results leave the device: the parked disposals of every thread go too
            var res = new Ciphertext[enc.Count];
            for (int i = 0; i < res.Length; i++)
                res[i] = SealInterop.ToCiphertext(words.Skip(i * dev.CtWords()).Take(dev.CtWords()).ToArray(), owner);
            return res;
        }
        /// <summary>dense plaintext blocks as device polynomials (N coefficients each); sparse constants as constant polynomials</summary>
        CnBuffer PlainOnDevice(AtomicSealBfvEncryptedEnvironment env)
        {
            if (plainDev != null) return plainDev;
            var dev = env.device;
            var buf = new CnBuffer(dev, (uint)plainData.Length, plain: true);
            var w = new ulong[plainData.Length * dev.N];
            for (int i = 0; i < plainData.Length; i++) Array.Copy(SealInterop.Coeffs(plainData[i], dev.N), 0, w, i * dev.N, dev.N);
            CnHip.Check(CnHip.cn_pt_upload(dev.Ctx, buf.Handle, 0, buf.Count, w));
            plainDev = buf;
            return buf;
        }

        // ---------------------------------------------------------------------------------------------------- HOT LOOP A
        static internal Task<AtomicSealBfvEncryptedVector> DenseMatrixBySparseVectorMultiplyTask(AtomicSealBfvEncryptedVector[] denses, AtomicSealBfvEncryptedVector sparse, AtomicSealBfvEncryptedEnvironment env)
        {
            return Task<AtomicSealBfvEncryptedVector>.Factory.StartNew(() => DenseMatrixBySparseVectorMultiply(denses, sparse, env));
        }
        /// <summary>out_block[i] = sum_k denses[k].block[i] * sparse[k]  (:434-521).  Encrypted columns x plain constants = ONE cn_scalar_dot per
        /// block (instead of K MultiplyPlain + K Add + AddMany): this is the call PoolLayer.ConvolveOnce issues per (map, corner)
        /// (PoolLayer.cs:113-121); with deferred submission the calls of a whole layer become one GEMM launch.</summary>
        static internal AtomicSealBfvEncryptedVector DenseMatrixBySparseVectorMultiply(AtomicSealBfvEncryptedVector[] denses, AtomicSealBfvEncryptedVector sparse, AtomicSealBfvEncryptedEnvironment env)
        {
            if ((ulong)denses.Length != sparse.Dim) throw new Exception("dimensions do not match");
            if (sparse.Format != EVectorFormat.sparse) throw new Exception("expecting a sparse vector");
            if (!denses[0].IsEncrypted && !sparse.IsEncrypted) throw new Exception("at least one parameter has to be encrypted");
            if (denses[0].IsSigned != sparse.IsSigned) throw new Exception("can't mix signed and unsigned messages");
            var ctx = env.device.Ctx;
            int l = (denses[0].enc != null) ? (int)denses[0].enc.Count : denses[0].plainData.Length;
            int K = denses.Length;
            var res = Result(env, (uint)l);
            res.Format = EVectorFormat.dense; res.Scale = denses[0].Scale * sparse.Scale; res.IsSigned = sparse.IsSigned; res.Dim = denses[0].Dim;
            if (denses[0].IsEncrypted && sparse.IsEncrypted)
            {   // Multiply + Relinearize per term (:459-465), then AddMany (:502)
                using (var terms = new CnBuffer(env.device, (uint)(K * l)))
                {
                    for (int k = 0; k < K; k++)
                        CnHip.Check(CnHip.cn_mul_relin(ctx, denses[k].enc.Handle, 0, 1, sparse.enc.Handle, (uint)k, 0, terms.Handle, (uint)(k * l), (uint)l));
                    for (int i = 0; i < l; i++)
                        CnHip.Check(CnHip.cn_add_many(ctx, terms.Handle, Enumerable.Range(0, K).Select(k => (uint)(k * l + i)).ToArray(), (uint)K, res.enc.Handle, (uint)i));
                }
                OperationsCount.Add(ref OperationsCount.Multiplication, K * l); OperationsCount.Add(ref OperationsCount.Relinarization, K * l);
            }
            else if (denses[0].IsEncrypted)
            {   // encrypted columns x plain constants (:466-474)
                var handles = denses.Select(d => d.enc.Handle).ToArray();
                var w = sparse.plainData.Select(SealInterop.Constant).ToArray();
                var index = new uint[K];
                for (int i = 0; i < l; i++)
                {
                    for (int k = 0; k < K; k++) index[k] = (uint)i;
                    CnHip.Check(CnHip.cn_scalar_dot(ctx, handles, index, w, (uint)K, res.enc.Handle, (uint)i));
                }
                res.enc.FreeNow = true;          // `using (conv = ConvolveOnce(..)) res[k] = conv.Add(bias)`: released at once, so that the queue can fold the bias addition
                OperationsCount.Add(ref OperationsCount.PlainMultiplication, w.Count(x => x != 0) * l);
            }
            else
            {   // plain dense columns x encrypted sparse entries (:476-485)
                using (var terms = new CnBuffer(env.device, (uint)K))
                    for (int i = 0; i < l; i++)
                    {
                        var used = new List<uint>();
                        for (int k = 0; k < K; k++)
                        {
                            if (denses[k].plainData[i].IsZero) continue;
                            CnHip.Check(CnHip.cn_mul_plain(ctx, sparse.enc.Handle, (uint)k, denses[k].PlainOnDevice(env).Handle, (uint)i, 1, terms.Handle, (uint)k, 1));
                            used.Add((uint)k);
                        }
                        CnHip.Check(CnHip.cn_add_many(ctx, terms.Handle, used.ToArray(), (uint)used.Count, res.enc.Handle, (uint)i));
                    }
            }
            OperationsCount.Add(ref OperationsCount.AddMany, l);
            return res;
        }

        internal Task<IVector> SparseMultiplyTask(IVector v, ulong colIndex, IComputationEnvironment env)
        {
            return Task<IVector>.Factory.StartNew(() => SparseMultiply(v, colIndex, env));
        }
        /// <summary>every block of this dense vector times ELEMENT colIndex of the sparse vector v (:529-598)</summary>
        internal IVector SparseMultiply(IVector v, ulong colIndex, IComputationEnvironment env)
        {
            if (colIndex >= v.Dim) throw new Exception("index exceeds dimension");
            var ev = v as AtomicSealBfvEncryptedVector;
            if (ev.Format != EVectorFormat.sparse) throw new Exception("expecting sparse format");
            if (ev.enc == null && this.enc == null) throw new Exception("at least one argument is expected to be encrypted");
            if (IsSigned != ev.IsSigned) throw new Exception("can't mix signed and unsigned numbers.");
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            var ctx = eenv.device.Ctx;
            uint n = (this.enc != null) ? this.enc.Count : (uint)this.plainData.Length;
            var t = Result(eenv, n);
            t.Scale = Scale * ev.Scale; t.Dim = Dim; t.IsSigned = IsSigned; t.Format = EVectorFormat.dense;
            if (this.enc != null && ev.enc != null)
            {
                CnHip.Check(CnHip.cn_mul_relin(ctx, ev.enc.Handle, (uint)colIndex, 0, this.enc.Handle, 0, 1, t.enc.Handle, 0, n));
                OperationsCount.Add(ref OperationsCount.Multiplication, (int)n); OperationsCount.Add(ref OperationsCount.Relinarization, (int)n);
                return t;
            }
            if (this.enc == null)
            {   // plain dense blocks x one encrypted element; a zero block becomes a fresh encryption of zero (:566)
                for (uint i = 0; i < n; i++)
                    if (plainData[i].IsZero) t.EncryptZeroInto(eenv, i);
                    else CnHip.Check(CnHip.cn_mul_plain(ctx, ev.enc.Handle, (uint)colIndex, PlainOnDevice(eenv).Handle, i, 1, t.enc.Handle, i, 1));
            }
            else
            {   // encrypted blocks x one plain constant (:580-596)
                if (ev.plainData[colIndex].IsZero) for (uint i = 0; i < n; i++) t.EncryptZeroInto(eenv, i);
                else CnHip.Check(CnHip.cn_mul_scalar(ctx, this.enc.Handle, 0, new ulong[] { SealInterop.Constant(ev.plainData[colIndex]) }, 0, t.enc.Handle, 0, n));
            }
            OperationsCount.Add(ref OperationsCount.PlainMultiplication, (int)n);
            return t;
        }
        /// <summary>a fresh encryption of zero into block `index`, made ON THE DEVICE with the public key (cn_encrypt, pt = 0): deferrable, no
        /// host encryption, no upload</summary>
        void EncryptZeroInto(AtomicSealBfvEncryptedEnvironment eenv, uint index)
        {
            CnHip.Check(CnHip.cn_encrypt(eenv.device.Ctx, 0, 0, 0, enc.Handle, index, 1, eenv.DeviceNonce()));
            OperationsCount.Add(ref OperationsCount.Encryption, 1);
        }

        // ---------------------------------------------------------------------------------------------------- packing (:600-761)
        /// <summary>Interleave / Stack: rotations, the row-boundary mask split and the AddMany of the lower / upper parts, as :600-722</summary>
        static CnBuffer Inteleave(AtomicSealBfvEncryptedVector[] vecs, int shift, int outputBlockCount, AtomicSealBfvEncryptedEnvironment env)
        {
            int blockSize = (int)env.builder.SlotCount;
            int absShift = (shift < 0) ? -shift : shift;
            if (shift < 0 && outputBlockCount > 1) throw new Exception("Negative shifts with multiple output blocks are not implemented yet");
            if ((absShift > blockSize / 2) && outputBlockCount > 1) throw new Exception("Shifts of more than half block size with multiple output blocks are not implemented yet");
            if (absShift * vecs.Length > blockSize * outputBlockCount) throw new Exception("not enough room for interleaving");
            var ctx = env.device.Ctx;
            int half = blockSize / 2;
            var lower = Enumerable.Range(0, outputBlockCount).Select(x => new List<uint>()).ToArray();
            var upper = Enumerable.Range(0, outputBlockCount).Select(x => new List<uint>()).ToArray();
            int n = vecs.Length;
            using (var work = new CnBuffer(env.device, (uint)(2 * n)))          // slot k: v, slot n + k: v2 (the split-off part)
            using (var tmp = new CnBuffer(env.device, 1))
            {
                // work[k] = RotateRows(vecs[k], steps[k]).  The reference copies and rotates vector by vector (:628-660); here the n copies are ONE launch
                // (cn_copy_many) and the n rotations ONE call whose hops run in rounds (cn_rotate_rows_many: every ciphertext with its own key and
                // element) - the same hops, the same words, the dispatches of the longest rotation instead of the sum
                var steps = new int[n]; var slot = new uint[n];
                for (int k = 0; k < n; k++)
                {
                    var thisShift = shift * k;
                    if (thisShift < 0) thisShift = half + thisShift;
                    var inBlockShift = thisShift % blockSize;
                    slot[k] = (uint)k;
                    if (inBlockShift == 0) steps[k] = 0;
                    else if (inBlockShift + absShift < half) steps[k] = -thisShift;
                    else if (inBlockShift >= half) steps[k] = -(inBlockShift - half);
                    else steps[k] = -inBlockShift;
                    if (steps[k] != 0) OperationsCount.Add(ref OperationsCount.Rotation, 1);
                }
                CnHip.Check(CnHip.cn_copy_many(ctx, vecs.Select(x => x.enc.Handle).ToArray(), new uint[n], (uint)n, work.Handle, 0));
                CnHip.Check(CnHip.cn_rotate_rows_many(ctx, work.Handle, slot, steps, (uint)n, work.Handle, slot));
                for (int k = 0; k < n; k++)
                {
                    var thisShift = shift * k;
                    if (thisShift < 0) thisShift = half + thisShift;
                    var inBlockShift = thisShift % blockSize;
                    var startBlock = thisShift / blockSize;
                    var endBlock = (thisShift + absShift) / blockSize;
                    uint v = (uint)k, v2 = (uint)(n + k);
                    if (inBlockShift == 0) lower[startBlock].Add(v);
                    else if (inBlockShift + absShift < half) lower[startBlock].Add(v);
                    else if (inBlockShift >= half)
                    {
                        if (startBlock == endBlock) upper[startBlock].Add(v);
                        else
                        {
                            SplitByMask(env, work, v, v2, inBlockShift + absShift - blockSize);
                            upper[startBlock].Add(v2); lower[endBlock].Add(v);
                        }
                    }
                    else
                    {
                        int upperPartSize = inBlockShift + absShift - half;
                        if (upperPartSize > 0)
                        {
                            SplitByMask(env, work, v, v2, upperPartSize);
                            upper[startBlock].Add(v); lower[startBlock].Add(v2);
                        }
                        else lower[startBlock].Add(v);
                    }
                }
                var res = new CnBuffer(env.device, (uint)outputBlockCount);
                for (int i = 0; i < outputBlockCount; i++)
                {
                    CnHip.Check(CnHip.cn_add_many(ctx, work.Handle, lower[i].ToArray(), (uint)lower[i].Count, res.Handle, (uint)i));
                    OperationsCount.Add(ref OperationsCount.AddMany, 1); OperationsCount.Add(ref OperationsCount.AddManyItemCount, lower[i].Count);
                    if (upper[i].Any())
                    {
                        CnHip.Check(CnHip.cn_add_many(ctx, work.Handle, upper[i].ToArray(), (uint)upper[i].Count, tmp.Handle, 0));
                        CnHip.Check(CnHip.cn_rotate_columns_add(ctx, tmp.Handle, 0, res.Handle, (uint)i, res.Handle, (uint)i, 1));      // RotateColumnsInplace + AddInplace
                        OperationsCount.Add(ref OperationsCount.AddMany, 1); OperationsCount.Add(ref OperationsCount.AddManyItemCount, upper[i].Count);
                        OperationsCount.Add(ref OperationsCount.Rotation, 1);
                    }
                }
                return res;
            }
        }
        /// <summary>v2 = v - v*ones(upperPartSize), v = v*ones(upperPartSize): the mask split of :664-676</summary>
        static void SplitByMask(AtomicSealBfvEncryptedEnvironment env, CnBuffer work, uint v, uint v2, int upperPartSize)
        {
            var ctx = env.device.Ctx;
            CnHip.Check(CnHip.cn_copy(ctx, work.Handle, v, work.Handle, v2, 1));
            using (var mask = new CnBuffer(env.device, 1, plain: true))
            {
                CnHip.Check(CnHip.cn_encode(ctx, Enumerable.Repeat(1UL, upperPartSize).ToArray(), (uint)upperPartSize, mask.Handle, 0));      // BatchEncoder.Encode(ones)
                CnHip.Check(CnHip.cn_mul_plain(ctx, work.Handle, v, mask.Handle, 0, 0, work.Handle, v, 1));
            }
            CnHip.Check(CnHip.cn_sub(ctx, work.Handle, v2, work.Handle, v, work.Handle, v2, 1));
            OperationsCount.Add(ref OperationsCount.PlainMultiplication, 1); OperationsCount.Add(ref OperationsCount.Subtraction, 1);
        }
        public static Task<AtomicSealBfvEncryptedVector> InterleaveTask(AtomicSealBfvEncryptedVector[] vecs, int shift, AtomicSealBfvEncryptedEnvironment env)
        {
            return Task<AtomicSealBfvEncryptedVector>.Factory.StartNew(() => Interleave(vecs, shift, env));
        }
        static public AtomicSealBfvEncryptedVector Interleave(AtomicSealBfvEncryptedVector[] vecs, int shift, AtomicSealBfvEncryptedEnvironment env)
        {
            if (vecs[0].Format != EVectorFormat.dense) throw new Exception("Expecting dense vector");
            var blockSize = env.builder.SlotCount;
            int outputBlocks = 1;
            if (shift > 0) outputBlocks = (int)Math.Ceiling(vecs[0].Dim * (ulong)vecs.Length / (double)blockSize);
            return new AtomicSealBfvEncryptedVector()
            {
                owner = env, enc = Inteleave(vecs, shift, outputBlocks, env), plainData = null, Dim = vecs[0].Dim, Scale = vecs[0].Scale,
                IsSigned = vecs[0].IsSigned, Format = EVectorFormat.dense
            };
        }
        static public Task<AtomicSealBfvEncryptedVector> StackTask(AtomicSealBfvEncryptedVector[] vecs, AtomicSealBfvEncryptedEnvironment env)
        {
            return Task<AtomicSealBfvEncryptedVector>.Factory.StartNew(() => Stack(vecs, env));
        }
        static public AtomicSealBfvEncryptedVector Stack(AtomicSealBfvEncryptedVector[] vecs, AtomicSealBfvEncryptedEnvironment env)
        {
            var res = Interleave(vecs, (int)vecs[0].Dim, env);
            res.Dim = vecs[0].Dim * (ulong)vecs.Length;
            return res;
        }

        // ---------------------------------------------------------------------------------------------------- HOT LOOP B (:774-860)
        public Task<IVector> PointwiseMultiplyTask(IVector v, IComputationEnvironment env) { return Task<IVector>.Factory.StartNew(() => PointwiseMultiply(v, env)); }
        /// <summary>one of the vectors is sparse of dimension 1: multiply every block of the other by that constant (:774-810)</summary>
        IVector PointwiseMultiplySparseDimOne(AtomicSealBfvEncryptedVector ev, AtomicSealBfvEncryptedEnvironment eenv)
        {
            var ctx = eenv.device.Ctx;
            uint n = (enc != null) ? enc.Count : (uint)plainData.Length;
            var t = Result(eenv, n);
            t.Scale = Scale * ev.Scale; t.Dim = Dim; t.Format = Format; t.IsSigned = IsSigned;
            if (this.enc != null && ev.enc != null)
            {
                CnHip.Check(CnHip.cn_mul_relin(ctx, ev.enc.Handle, 0, 0, enc.Handle, 0, 1, t.enc.Handle, 0, n));
                OperationsCount.Add(ref OperationsCount.Multiplication, (int)n); OperationsCount.Add(ref OperationsCount.Relinarization, (int)n);
                return t;
            }
            if (this.enc != null) CnHip.Check(CnHip.cn_mul_scalar(ctx, enc.Handle, 0, new ulong[] { SealInterop.Constant(ev.plainData[0]) }, 0, t.enc.Handle, 0, n));
            else for (uint i = 0; i < n; i++) CnHip.Check(CnHip.cn_mul_plain(ctx, ev.enc.Handle, 0, PlainOnDevice(eenv).Handle, i, 1, t.enc.Handle, i, 1));
            OperationsCount.Add(ref OperationsCount.PlainMultiplication, (int)n);
            return t;
        }
        /// <summary>per block Multiply + Relinearize (:839-840) = cn_mul_relin over all blocks; ct x pt = MultiplyPlain (:855).  SquareActivation
        /// calls this once per column (EncryptedSealBfvMatrix.cs:140-154): deferred submission merges the columns of a layer into one batch.</summary>
        public IVector PointwiseMultiply(IVector v, IComputationEnvironment env)
        {
            var ev = v as AtomicSealBfvEncryptedVector;
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            if (IsSigned != ev.IsSigned) throw new Exception("Can't mix signed and unsigned numbers.");
            if (this.plainData != null && ev.plainData != null) throw new Exception("multiplying two plaintexts is not implemented");
            if (Dim == 1 && Format == EVectorFormat.sparse) return ev.PointwiseMultiplySparseDimOne(this, eenv);
            if (ev.Dim == 1 && ev.Format == EVectorFormat.sparse) return PointwiseMultiplySparseDimOne(ev, eenv);
            if (Dim != v.Dim) throw new Exception("Dimensions do not match");
            if (Format != ev.Format) throw new Exception("Format mismatch");
            var ctx = eenv.device.Ctx;
            if (this.enc != null && ev.enc != null)
            {
                var t = Result(eenv, ev.enc.Count);
                t.Scale = Scale * ev.Scale; t.Dim = Dim; t.Format = Format; t.IsSigned = IsSigned;
                CnHip.Check(CnHip.cn_mul_relin(ctx, ev.enc.Handle, 0, 1, enc.Handle, 0, 1, t.enc.Handle, 0, t.enc.Count));
                OperationsCount.Add(ref OperationsCount.Multiplication, (int)t.enc.Count); OperationsCount.Add(ref OperationsCount.Relinarization, (int)t.enc.Count);
                return t;
            }
            var e = enc ?? ev.enc;
            var pl = (enc == null) ? this : ev;
            var r = Result(eenv, e.Count);
            r.Scale = Scale * ev.Scale; r.Dim = Dim; r.Format = Format; r.IsSigned = IsSigned;
            if (Format == EVectorFormat.dense) CnHip.Check(CnHip.cn_mul_plain(ctx, e.Handle, 0, pl.PlainOnDevice(eenv).Handle, 0, 1, r.enc.Handle, 0, e.Count));
            else CnHip.Check(CnHip.cn_mul_scalar(ctx, e.Handle, 0, pl.plainData.Select(SealInterop.Constant).ToArray(), 1, r.enc.Handle, 0, e.Count));
            OperationsCount.Add(ref OperationsCount.PlainMultiplication, (int)e.Count);
            return r;
        }

        // ---------------------------------------------------------------------------------------------------- HOT LOOP C (:862-977)
        public Task<IVector> SumAllSlotsTask(IComputationEnvironment env) { return Task<IVector>.Factory.StartNew(() => SumAllSlots(Int32.MaxValue, env)); }
        public IVector SumAllSlots(IComputationEnvironment env) => SumAllSlots(env, null);
        public IVector SumAllSlots(IComputationEnvironment env, int? ForceOutputInColumn = null) { return SumAllSlots(Int32.MaxValue, env, ForceOutputInColumn); }
        public Task<IVector> SumAllSlotsTask(ulong length, IComputationEnvironment env) { return Task<IVector>.Factory.StartNew(() => SumAllSlots(length, env)); }
        public IVector SumAllSlots(ulong length, IComputationEnvironment env) => SumAllSlots(length, env, null);
        /// <summary>AddMany of the blocks, column swap + add when length >= N/2, log2 rotate-and-add steps (cn_sum_slots), optional one-hot mask</summary>
        public IVector SumAllSlots(ulong length, IComputationEnvironment env, int? ForceOutputInColumn = null) { return SumAllSlots(length, env, ForceOutputInColumn, false); }
        /// <summary>consume: the caller owns this vector as a temporary (the product inside DotProduct) - the sum is built in its array instead of
        /// in a copy; this object is spent afterwards</summary>
        IVector SumAllSlots(ulong length, IComputationEnvironment env, int? ForceOutputInColumn, bool consume)
        {
            if (Format != EVectorFormat.dense) throw new Exception("Expecting dense vector format");
            if (length != Int32.MaxValue && ForceOutputInColumn != null) throw new Exception("forcing output in a column works only when doing complete sum");
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            if (plainData != null) throw new Exception("SumAllSlots can be applied to encrypted data only");
            if (length <= 0) throw new Exception("Can't sum over less then one element");
            if (length == 1) return this;
            var ctx = eenv.device.Ctx;
            ulong slots = eenv.builder.SlotCount;
            CnBuffer sum;
            if (consume && enc.Count == 1) { sum = enc; enc = null; }
            else
            {
                sum = new CnBuffer(eenv.device, 1);
                if (enc.Count > 1)
                {
                    CnHip.Check(CnHip.cn_add_many(ctx, enc.Handle, Enumerable.Range(0, (int)enc.Count).Select(i => (uint)i).ToArray(), enc.Count, sum.Handle, 0));
                    OperationsCount.Add(ref OperationsCount.AddMany, 1); OperationsCount.Add(ref OperationsCount.AddManyItemCount, (int)enc.Count);
                }
                else CnHip.Check(CnHip.cn_copy(ctx, enc.Handle, 0, sum.Handle, 0, 1));
                if (consume) Dispose();
            }
            CnHip.Check(CnHip.cn_sum_slots(ctx, sum.Handle, 0, 1, length >= slots ? 0 : (uint)length));      // RotateColumns + Add, then RotateRows(-2^s) + AddInplace (:914-930)
            if (length >= slots / 2) length = slots / 2;
            if (ForceOutputInColumn != null)
            {
                int col = ForceOutputInColumn.Value;
                using (var mask = new CnBuffer(eenv.device, 1, plain: true))
                {
                    var onehot = new ulong[col + 1]; onehot[col] = 1;
                    CnHip.Check(CnHip.cn_encode(ctx, onehot, (uint)onehot.Length, mask.Handle, 0));
                    CnHip.Check(CnHip.cn_mul_plain(ctx, sum.Handle, 0, mask.Handle, 0, 0, sum.Handle, 0, 1));
                }
                length = 1;
            }
            return new AtomicSealBfvEncryptedVector()
            {
                owner = eenv, IsSigned = IsSigned, Scale = Scale, Dim = (length >= slots / 2) ? 1 : this.Dim, enc = sum, plainData = null,
                Format = (length >= slots) ? EVectorFormat.sparse : EVectorFormat.dense
            };
        }
        public Task<IVector> DotProductTask(IVector v, IComputationEnvironment env, int? ForceOutputInColumn = null) { return Task<IVector>.Factory.StartNew(() => DotProduct(v, env, ForceOutputInColumn)); }
        public IVector DotProduct(IVector v, IComputationEnvironment env) => DotProduct(v, env, null);
        public IVector DotProduct(IVector v, IComputationEnvironment env, int? ForceOutputInColumn = null)
        {
            var mul = (AtomicSealBfvEncryptedVector)PointwiseMultiply(v, env);          // a temporary: summed in its own array
            return mul.SumAllSlots(Int32.MaxValue, env, ForceOutputInColumn, true);
        }
        public Task<IVector> DotProductTask(IVector v, ulong length, IComputationEnvironment env) { return Task<IVector>.Factory.StartNew(() => DotProduct(v, length, env)); }
        public IVector DotProduct(IVector v, ulong length, IComputationEnvironment env)
        {
            var mul = (AtomicSealBfvEncryptedVector)PointwiseMultiply(v, env);
            if (length == 1) return mul;
            return mul.SumAllSlots(length, env, null, true);
        }

        // ---------------------------------------------------------------------------------------------------- linear (:983-1024, 1238-1271)
        public Task<IVector> AddTask(IVector v, IComputationEnvironment env) { return Task<IVector>.Factory.StartNew(() => Add(v, env)); }
        public IVector Add(IVector v, IComputationEnvironment env)
        {
            if (Scale == 0) return v;
            if (v.Scale == 0) return this;
            if (Scale != v.Scale) throw new Exception("Scales do not match.");
            if (Dim != v.Dim) throw new Exception("Dimensions do not match");
            var ev = v as AtomicSealBfvEncryptedVector;
            if (Format != ev.Format) throw new Exception("Format mismatch");
            if (IsSigned != ev.IsSigned) throw new Exception("can't mix signed and unsigned numbers.");
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            if (this.plainData != null && ev.plainData != null) throw new Exception("adding two plaintexts is not supported");
            var ctx = eenv.device.Ctx;
            if (this.enc != null && ev.enc != null)
            {
                var t = Result(eenv, ev.enc.Count);
                t.Scale = Scale; t.Dim = Dim; t.Format = Format; t.IsSigned = IsSigned;
                CnHip.Check(CnHip.cn_add(ctx, ev.enc.Handle, 0, enc.Handle, 0, t.enc.Handle, 0, t.enc.Count));
                OperationsCount.Add(ref OperationsCount.Addition, (int)t.enc.Count);
                return t;
            }
            var e = enc ?? ev.enc;
            var pl = (enc == null) ? this : ev;
            var r = Result(eenv, e.Count);
            r.Scale = Scale; r.Dim = Dim; r.Format = Format; r.IsSigned = IsSigned;
            CnHip.Check(CnHip.cn_add_plain(ctx, e.Handle, 0, pl.PlainOnDevice(eenv).Handle, 0, 0, r.enc.Handle, 0, (uint)pl.plainData.Length));      // Evaluator.AddPlain (:1019)
            OperationsCount.Add(ref OperationsCount.PlainAddition, pl.plainData.Length);
            return r;
        }
        public Task<IVector> SubtractTask(IVector v, IComputationEnvironment env) { return Task<IVector>.Factory.StartNew(() => Subtract(v, env)); }
        public IVector Subtract(IVector v, IComputationEnvironment env)
        {
            if (v.Scale == 0) return this;
            if (Scale != v.Scale) throw new Exception("Scales do not match.");
            if (Dim != v.Dim) throw new Exception("Dimensions do not match");
            var ev = v as AtomicSealBfvEncryptedVector;
            if (Format != ev.Format) throw new Exception("Format mismatch");
            if (IsSigned != ev.IsSigned) throw new Exception("Can't mix signed and unsigned numbers.");
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            if (this.plainData != null) throw new Exception("the first argument for subtraction must be encrypted");
            var ctx = eenv.device.Ctx;
            var t = Result(eenv, enc.Count);
            t.Scale = Scale; t.Dim = Dim; t.Format = Format; t.IsSigned = IsSigned;
            if (ev.enc != null)
            {
                CnHip.Check(CnHip.cn_sub(ctx, enc.Handle, 0, ev.enc.Handle, 0, t.enc.Handle, 0, enc.Count));
                OperationsCount.Add(ref OperationsCount.Subtraction, (int)enc.Count);
                return t;
            }
            CnHip.Check(CnHip.cn_add_plain(ctx, enc.Handle, 0, ev.PlainOnDevice(eenv).Handle, 0, 1, t.enc.Handle, 0, enc.Count));              // Evaluator.SubPlain (:1267)
            OperationsCount.Add(ref OperationsCount.PlainSubtraction, (int)enc.Count);
            return t;
        }

        // ---------------------------------------------------------------------------------------------------- decryption: client side, SEAL (:1030-1110)
        public Task<Vector<double>> DecryptTask(IComputationEnvironment env) { return Task<Vector<double>>.Factory.StartNew(() => Decrypt(env)); }
        IEnumerable<ulong> DecryptResidues(AtomicSealBfvEncryptedEnvironment eenv)
        {
            var res = new List<ulong>();
            Ciphertext[] cts = (enc != null) ? Download() : null;
            int length = (enc == null) ? plainData.Length : cts.Length;
            var plain = new Plaintext(eenv.memoryPool);
            for (int i = 0; i < length; i++)
            {
                if (cts != null)
                {
                    CryptoTracker.TestBudget(cts[i], eenv.decryptor);
                    eenv.decryptor.Decrypt(cts[i], plain);
                    OperationsCount.Add(ref OperationsCount.Decryption, 1);
                    cts[i].Dispose();
                }
                else plain = plainData[i];
                if (Format == EVectorFormat.dense)
                {
                    List<ulong> local = new List<ulong>();
                    eenv.builder.Decode(plain, local);
                    int left = (int)Dim - res.Count;
                    res.AddRange(local.Take((left > local.Count) ? local.Count : left));
                }
                else res.Add(SealInterop.Constant(plain));
            }
            return res;
        }
        public Vector<double> Decrypt(IComputationEnvironment env)
        {
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            var mod = (double)eenv.parameters.PlainModulus.Value;
            return Vector<double>.Build.DenseOfEnumerable(DecryptResidues(eenv)
                .Select(v => ((IsSigned && v * 2 > eenv.parameters.PlainModulus.Value) ? v - mod : v) / Scale));
        }
        public Task<IEnumerable<BigInteger>> DecryptFullPrecisionTask(IComputationEnvironment env) { return Task<IEnumerable<BigInteger>>.Factory.StartNew(() => DecryptFullPrecision(env)); }
        public IEnumerable<BigInteger> DecryptFullPrecision(IComputationEnvironment env)
        {
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            var mod = new BigInteger(eenv.parameters.PlainModulus.Value);
            return DecryptResidues(eenv).Select(v => (IsSigned && v * 2 > eenv.parameters.PlainModulus.Value) ? new BigInteger(v) - mod : new BigInteger(v)).ToList();
        }

        // ---------------------------------------------------------------------------------------------------- encoding / encryption: client side, SEAL (:1114-1232)
        Plaintext[] VectorToPlaintext(Vector<double> v, AtomicSealBfvEncryptedEnvironment eenv)
        {
            if (Scale == 0) Scale = 1;
            var values = v.Multiply(Scale).PointwiseRound().Select(x => (ulong)((!IsSigned || x >= 0) ? x : eenv.parameters.PlainModulus.Value + x)).ToArray();
            return VectorToPlaintext(values, eenv);
        }
        Plaintext[] VectorToPlaintext(UInt64[] v, AtomicSealBfvEncryptedEnvironment eenv)
        {
            var lst = new List<Plaintext>();
            int slots = (int)eenv.builder.SlotCount;
            for (int start = 0; start < v.Length;)
            {
                if (Format == EVectorFormat.dense)
                {
                    int size = (start + slots <= v.Length) ? slots : v.Length - start;
                    var p = new Plaintext(eenv.memoryPool);
                    eenv.builder.Encode(v.Skip(start).Take(size).ToList(), p);
                    lst.Add(p);
                    start += size;
                }
                else
                {
                    lst.Add(new Plaintext(v[start].ToString("X"), eenv.memoryPool));
                    start++;
                }
            }
            return lst.ToArray();
        }
        Plaintext DoubleToPlaintext(double v, AtomicSealBfvEncryptedEnvironment eenv)
        {
            if (Scale == 0) Scale = 1;
            var value = Math.Round(v * Scale);
            var unsigned = (ulong)((!IsSigned || value >= 0) ? value : eenv.parameters.PlainModulus.Value + value);
            return new Plaintext(unsigned.ToString("X"), eenv.memoryPool);
        }
        void Plain(Vector<double> v, EVectorFormat Format, IComputationEnvironment env)
        {
            OperationsCount.Add(ref OperationsCount.Plain, 1);
            owner = env as AtomicSealBfvEncryptedEnvironment; this.Format = Format;
            plainData = VectorToPlaintext(v, owner); enc = null; Dim = (ulong)v.Count;
        }
        void Plain(UInt64[] v, EVectorFormat Format, IComputationEnvironment env)
        {
            OperationsCount.Add(ref OperationsCount.Plain, 1);
            owner = env as AtomicSealBfvEncryptedEnvironment; this.Format = Format;
            plainData = VectorToPlaintext(v, owner); enc = null; Dim = (ulong)v.Length;
        }
        /// <summary>Encryptor.Encrypt per plaintext on the client (:1211), ONE upload of all blocks to the device</summary>
        void EncryptPlaintexts(Plaintext[] plain, AtomicSealBfvEncryptedEnvironment eenv)
        {
            owner = eenv;
            if (plain.Length == 1 && plain[0].IsZero)
            {
                // the zero vector of a padded convolution tap (PoolLayer.ElementAt: Factory.GetEncryptedVector(zeros), PoolLayer.cs:67-80) - 645 per layer and
                // plaintext prime in CryptoNets: AllocateCiphertext + Encrypt(PlainZero) as ONE library call (cn_encrypt_zero_new), queued like the evaluator calls
                ulong zh;
                CnHip.Check(CnHip.cn_encrypt_zero_new(eenv.device.Ctx, eenv.DeviceNonce(), out zh));
                enc = new CnBuffer(eenv.device, zh, 1);
                plain[0].Dispose();
                plainData = null;
                OperationsCount.Add(ref OperationsCount.Encryption, 1);
                return;
            }
            enc = new CnBuffer(eenv.device, (uint)plain.Length);
            if (plain.All(p => p.IsZero))
            {
                // an all-zero vector of several blocks: nothing secret goes in, the public key is on the device - one queued cn_encrypt instead of SEAL
                // Encrypt + upload per block
                CnHip.Check(CnHip.cn_encrypt(eenv.device.Ctx, 0, 0, 0, enc.Handle, 0, enc.Count, eenv.DeviceNonce()));
                foreach (var p in plain) p.Dispose();
                plainData = null;
                OperationsCount.Add(ref OperationsCount.Encryption, plain.Length);
                return;
            }
            int ctw = eenv.device.CtWords();
            var words = new ulong[plain.Length * ctw];
            using (var c = new Ciphertext(eenv.context, eenv.memoryPool))
                for (int i = 0; i < plain.Length; i++)
                {
                    eenv.encryptor.Encrypt(plain[i], c, eenv.memoryPool);
                    Array.Copy(SealInterop.Words(c), 0, words, i * ctw, ctw);
                    plain[i].Dispose();
                }
            CnHip.Check(CnHip.cn_ct_upload(eenv.device.Ctx, enc.Handle, 0, enc.Count, words));
            plainData = null;
            OperationsCount.Add(ref OperationsCount.Encryption, plain.Length);
        }
        void Encrypt(Vector<double> v, EVectorFormat format, IComputationEnvironment env)
        {
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            this.Format = format;
            EncryptPlaintexts(VectorToPlaintext(v, eenv), eenv);
            Dim = (ulong)v.Count;
        }
        void Encrypt(UInt64[] v, EVectorFormat format, IComputationEnvironment env)
        {
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            this.Format = format;
            EncryptPlaintexts(VectorToPlaintext(v, eenv), eenv);
            Dim = (ulong)v.Length;
        }

        // ---------------------------------------------------------------------------------------------------- persistence (:1273-1345): SEAL object streams
        public void Write(StreamWriter str)
        {
            str.WriteLine("<Start EncryptedVector>");
            str.WriteLine(Scale); str.WriteLine(IsSigned); str.WriteLine(Enum.GetName(Format.GetType(), Format)); str.WriteLine(Dim);
            using (MemoryStream mem = new MemoryStream())
            {
                if (plainData == null)
                {
                    var cts = Download();
                    str.WriteLine("Encrypted"); str.WriteLine(cts.Length);
                    foreach (var c in cts) { c.Save(mem); c.Dispose(); }
                }
                else
                {
                    str.WriteLine("Plain"); str.WriteLine(plainData.Length);
                    for (int i = 0; i < plainData.Length; i++) plainData[i].Save(mem);
                }
                mem.Flush(); mem.Position = 0;
                str.WriteLine(Convert.ToBase64String(mem.ToArray(), Base64FormattingOptions.None));
                str.WriteLine("<End EncryptedVector>");
                str.Flush();
            }
        }
        public static AtomicSealBfvEncryptedVector Read(StreamReader str, AtomicSealBfvEncryptedEnvironment env)
        {
            var vct = new AtomicSealBfvEncryptedVector() { owner = env };
            if (str.ReadLine() != "<Start EncryptedVector>") throw new Exception("Bad stream format.");
            vct.Scale = Double.Parse(str.ReadLine());
            vct.IsSigned = Boolean.Parse(str.ReadLine());
            vct.Format = (EVectorFormat)Enum.Parse(vct.Format.GetType(), str.ReadLine());
            vct.Dim = ulong.Parse(str.ReadLine());
            var mode = str.ReadLine();
            var length = int.Parse(str.ReadLine());
            using (var mem = new MemoryStream(Convert.FromBase64String(str.ReadLine())))
            {
                switch (mode)
                {
                    case "Encrypted":
                        vct.enc = new CnBuffer(env.device, (uint)length);
                        using (var c = new Ciphertext(env.context, env.memoryPool))
                            for (int i = 0; i < length; i++)
                            {
                                c.Load(env.context, mem);
                                CnHip.Check(CnHip.cn_ct_upload(env.device.Ctx, vct.enc.Handle, (uint)i, 1, SealInterop.Words(c)));
                            }
                        break;
                    case "Plain":
                        vct.plainData = new Plaintext[length];
                        for (int i = 0; i < length; i++) { vct.plainData[i] = new Plaintext(env.memoryPool); vct.plainData[i].Load(env.context, mem); }
                        break;
                    default: throw new Exception("unknown format");
                }
            }
            if (str.ReadLine() != "<End EncryptedVector>") throw new Exception("Bad stream format.");
            return vct;
        }

        // ---------------------------------------------------------------------------------------------------- misc (:1347-1475)
        public static AtomicSealBfvEncryptedVector GenerateSparseOfArray(AtomicSealBfvEncryptedVector[] encryptedVector, IComputationEnvironment env)
        {
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            var res = Result(eenv, (uint)encryptedVector.Length);
            res.Scale = encryptedVector[0].Scale; res.Dim = (ulong)encryptedVector.Length; res.Format = EVectorFormat.sparse; res.IsSigned = encryptedVector[0].IsSigned;
            // the first block of every vector into one sparse vector: one launch (the reference copies element by element, :1352-1357)
            CnHip.Check(CnHip.cn_copy_many(eenv.device.Ctx, encryptedVector.Select(x => x.enc.Handle).ToArray(), new uint[encryptedVector.Length],
                                           (uint)encryptedVector.Length, res.enc.Handle, 0));
            return res;
        }
        public void RegisterScale(double scale) { Scale = scale; }

        public Task<IVector> DuplicateTask(ulong count, IComputationEnvironment env) { return Task<IVector>.Factory.StartNew(() => Duplicate(count, env)); }
        public IVector Duplicate(ulong count, IComputationEnvironment env)
        {
            ulong shift = 1;
            while (shift < Dim) shift *= 2;
            if (enc == null) throw new Exception("Duplicate operates only on encrypted data");
            if (Format == EVectorFormat.sparse) throw new Exception("Duplicate operates only on dense vectors");
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            int slots = (int)eenv.builder.SlotCount;
            if (shift * count > (ulong)slots) throw new Exception("Packed vector must fit in a single ciphertext");
            var ctx = eenv.device.Ctx;
            var res = new CnBuffer(eenv.device, 1);
            // every copy is a rotation of THIS ciphertext (or of its column-swapped form) and they are only added up (RotateRowsAndAdd into the
            // result, :862-868, :1385-1404): the count - 1 rotations are one call whose hops share rounds (cn_rotate_rows_many), the sum one
            // AddMany - modular addition is exact in any order, so these are the words of the reference's rotate-and-add chain
            int n = (int)count - 1;
            if (n <= 0) { CnHip.Check(CnHip.cn_copy(ctx, enc.Handle, 0, res.Handle, 0, 1)); }
            else using (var work = new CnBuffer(eenv.device, (uint)(2 + n)))          // 0: this vector, 1: its column-swapped form, 2 ..: the rotated copies
            {
                CnHip.Check(CnHip.cn_copy(ctx, enc.Handle, 0, work.Handle, 0, 1));
                var src = new uint[n]; var steps = new int[n]; var dst = new uint[n]; var terms = new uint[n + 1];
                bool columnRotated = false;
                for (int i = 1; i <= n; i++)
                {
                    int targetShiftSize = (int)((ulong)i * shift);
                    if (targetShiftSize * 2 >= slots) { columnRotated = true; src[i - 1] = 1; targetShiftSize -= slots / 2; }
                    steps[i - 1] = -targetShiftSize; dst[i - 1] = (uint)(1 + i); terms[i] = (uint)(1 + i);
                    OperationsCount.Add(ref OperationsCount.Rotation, 1); OperationsCount.Add(ref OperationsCount.Addition, 1);
                }
                if (columnRotated)
                {
                    CnHip.Check(CnHip.cn_rotate_columns(ctx, enc.Handle, 0, work.Handle, 1, 1));
                    OperationsCount.Add(ref OperationsCount.Rotation, 1);
                }
                CnHip.Check(CnHip.cn_rotate_rows_many(ctx, work.Handle, src, steps, (uint)n, work.Handle, dst));
                CnHip.Check(CnHip.cn_add_many(ctx, work.Handle, terms, (uint)(n + 1), res.Handle, 0));
            }
            return new AtomicSealBfvEncryptedVector() { owner = eenv, IsSigned = IsSigned, Scale = Scale, Dim = count * shift, enc = res, plainData = null, Format = EVectorFormat.dense };
        }
        public Task<IVector> RotateTask(int amount, IComputationEnvironment env) { return Task<IVector>.Factory.StartNew(() => Rotate(amount, env)); }
        public IVector Rotate(int amount, IComputationEnvironment env)
        {
            if (enc == null) throw new Exception("Rotate operates only on encrypted data");
            if (Format == EVectorFormat.sparse) throw new Exception("Rotate operates only on dense vectors");
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            var res = new CnBuffer(eenv.device, 1);
            CnHip.Check(CnHip.cn_rotate_rows(eenv.device.Ctx, enc.Handle, 0, amount, res.Handle, 0, 1));
            return new AtomicSealBfvEncryptedVector() { owner = eenv, IsSigned = IsSigned, Scale = Scale, Dim = Dim, enc = res, plainData = null, Format = EVectorFormat.dense };
        }
        internal Task<IVector> PermuteTask(IVector[] selections, int[] shifts, ulong outputDim, IComputationEnvironment env)
        {
            return Task<IVector>.Factory.StartNew(() => Permute(selections, shifts, outputDim, env));
        }
        /// <summary>sum_i RotateRows(x * selection_i, shifts[i]) (:1436-1475); the additions ride on the rotations (cn_rotate_rows_add)</summary>
        public IVector Permute(IVector[] selections, int[] shifts, ulong outputDim, IComputationEnvironment env)
        {
            if (Format != EVectorFormat.dense) throw new Exception("Permute works only on dense vectors");
            if (selections.Length != shifts.Length) throw new Exception("number of selection vectors and number of shifts does not match");
            if (plainData != null) throw new Exception("can permute only encrypted vectors");
            if (enc.Count > 1) throw new Exception("can permute only a single block");
            var eenv = env as AtomicSealBfvEncryptedEnvironment;
            var ctx = eenv.device.Ctx;
            CnBuffer res = null;
            int first = -1;
            using (var t = new CnBuffer(eenv.device, 1))
                for (int i = 0; i < selections.Length; i++)
                {
                    if (selections[i] == null) continue;
                    if (first < 0) first = i;
                    if (selections[i].Dim != Dim) throw new Exception("dimension of selection vector does not match dimension of data vector");
                    if (selections[i].Scale != selections[first].Scale) throw new Exception("scales of all selection vectors should be the same");
                    var s = selections[i] as AtomicSealBfvEncryptedVector;
                    if (s.plainData != null) CnHip.Check(CnHip.cn_mul_plain(ctx, enc.Handle, 0, s.PlainOnDevice(eenv).Handle, 0, 1, t.Handle, 0, 1));
                    else CnHip.Check(CnHip.cn_mul_relin(ctx, enc.Handle, 0, 1, s.enc.Handle, 0, 1, t.Handle, 0, 1));      // (the reference multiplies without relinearising, :1457, and then cannot rotate)
                    if (res == null)
                    {
                        res = new CnBuffer(eenv.device, 1);
                        CnHip.Check(CnHip.cn_rotate_rows(ctx, t.Handle, 0, shifts[i], res.Handle, 0, 1));
                    }
                    else CnHip.Check(CnHip.cn_rotate_rows_add(ctx, t.Handle, 0, shifts[i], res.Handle, 0, res.Handle, 0, 1));
                }
            if (first < 0) throw new Exception("permuting with no selected values is illigal");
            return new AtomicSealBfvEncryptedVector()
            {
                owner = eenv, IsSigned = IsSigned, Scale = Scale * selections[first].Scale, Dim = outputDim, enc = res, plainData = null, Format = EVectorFormat.dense
            };
        }
    }
}
#endif
