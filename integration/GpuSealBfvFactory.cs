// GpuSealBfvFactory: the IFactory a CryptoNets / LoLa program instantiates instead of EncryptedSealBfvFactory to run on MI355X.
//
// With CNHIP defined, `EncryptedSealBfvFactory` (IFactory.cs:240-410, unchanged) already builds its environments from the GPU twin of
// AtomicSealBfvEncryptedEnvironment (GpuAtomicSealBfvEncryptedVector.cs) - its environment pool (IFactory.cs:281-294), vector / matrix
// constructors and CRT helpers work as they are.  This subclass only adds what a GPU deployment has to choose:
//   * which device(s) the plaintext-prime channels live on (one context per prime; round-robin over `devices`),
//   * deferred submission (on by default: the layers' per-ciphertext calls are merged into batched launches by libcnhip),
//   * optionally a cap on the caller thread count.  The reference uses Environment.ProcessorCount threads per ParallelProcessInEnv
//     (Defaults.cs) to keep CPU cores busy with SEAL arithmetic; here the threads only enqueue work.  libcnhip does not need the cap -
//     its context lock lets at most three waiters spin and puts the others to sleep on a futex (cn_host.cpp; the owner bias and the
//     combining experiments of round 3 ship OFF: CN_LOCK_GRACE_NS = 0, CN_LOCK_COMBINE = 0).  Measured: the unchanged caller runs at
//     0.82-0.92 of the batched rate from 1 to 256 caller threads (profiles/r03_unchanged_caller_lock.txt) - not flat, but without a cliff -
//     so a program that keeps `new EncryptedSealBfvFactory(...)` and the default thread count works; the cap (callerThreads > 0) saves the
//     program its own thread start-up and the last ~10 %,
//   * replicas on other GPUs for independent batches: same SEAL keys, evaluation keys copied with ONE RCCL broadcast per key over xGMI
//     (cn_ctx_broadcast_keys) - no ciphertext ever crosses GPUs (SURVEY.md section 8e).
// Not compiled in this repository (no .NET toolchain in the build image); see INTEGRATION.md.
#if CNHIP
using System;
using System.IO;
using System.Linq;

namespace HEWrapper
{
    public class GpuSealBfvFactory : EncryptedSealBfvFactory
    {
        /// <summary>threads that issue evaluator calls: 0 = leave Defaults.ThreadCount as the program set it (the reference's default is
        /// Environment.ProcessorCount); the library takes any count</summary>
        public const int DefaultCallerThreads = 0;

        public static int DeviceCount { get { return CnHip.cn_device_count(); } }

        static void Configure(int device, bool deferred, int callerThreads)
        {
            if (CnHip.cn_device_count() <= 0) throw new Exception("no MI355X visible to libcnhip (there is no CPU fallback)");
            AtomicSealBfvEncryptedEnvironment.DefaultDeviceIndex = device;
            AtomicSealBfvEncryptedEnvironment.DeferredSubmission = deferred;
            if (callerThreads > 0 && Defaults.ThreadCount > callerThreads) Defaults.ThreadCount = callerThreads;
        }
        // the base constructors generate the keys (SEAL KeyGenerator) and, through SetKeys, create the device contexts and upload the
        // evaluation keys; the static configuration above must therefore be in place BEFORE the base constructor runs
        static ulong[] Configured(ulong[] primes, int device, bool deferred, int callerThreads) { Configure(device, deferred, callerThreads); return primes; }
        static string Configured(string fileName, int device, bool deferred, int callerThreads) { Configure(device, deferred, callerThreads); return fileName; }
        static Stream Configured(Stream stream, int device, bool deferred, int callerThreads) { Configure(device, deferred, callerThreads); return stream; }

        /// <summary>new keys; all plaintext-prime channels on GPU `device`</summary>
        public GpuSealBfvFactory(ulong[] primes, ulong n, int DecompositionBitCount = 10, int GaloisDecompositionBitCount = 20, int SmallModulusCount = -1,
                                 int device = 0, bool deferredSubmission = true, int callerThreads = DefaultCallerThreads)
            : base(Configured(primes, device, deferredSubmission, callerThreads), n, DecompositionBitCount, GaloisDecompositionBitCount, SmallModulusCount) { }

        /// <summary>the reference's default parameters (IFactory.cs:247-253): N = 4096, five plaintext primes</summary>
        public GpuSealBfvFactory(int device = 0, bool deferredSubmission = true, int callerThreads = DefaultCallerThreads)
            : base(Configured(new ulong[] { 40961, 65537, 114689, 147457, 188417 }, device, deferredSubmission, callerThreads), 4096) { }

        /// <summary>keys from the zip container EncryptedSealBfvFactory.Save wrote (IFactory.cs:262-276)</summary>
        public GpuSealBfvFactory(string fileName, int device = 0, bool deferredSubmission = true, int callerThreads = DefaultCallerThreads)
            : base(Configured(fileName, device, deferredSubmission, callerThreads)) { }
        public GpuSealBfvFactory(Stream stream, int device = 0, bool deferredSubmission = true, int callerThreads = DefaultCallerThreads)
            : base(Configured(stream, device, deferredSubmission, callerThreads)) { }

        EncryptedSealBfvEnvironment Reference { get { return (EncryptedSealBfvEnvironment)AllocateComputationEnv(); } }

        /// <summary>waits for everything queued on the factory's device contexts (deferred calls are launched first)</summary>
        public void Synchronize()
        {
            var env = Reference;
            foreach (var e in env.Environments) CnHip.Check(CnHip.cn_sync(e.device.Ctx));
            FreeComputationEnv(env);
        }

        /// <summary>A replica of this factory's evaluation side on another GPU, for an independent batch: the SAME SEAL objects (keys,
        /// encoder, encryptor, decryptor - client side) with fresh device contexts on `device`, whose relinearisation / Galois keys arrive
        /// through cn_ctx_broadcast_keys (RCCL over xGMI).  Batches evaluated on replicas never exchange ciphertexts.</summary>
        public AtomicSealBfvEncryptedEnvironment[] ReplicateTo(int device)
        {
            var env = Reference;
            var res = env.Environments.Select(src =>
            {
                var dst = new AtomicSealBfvEncryptedEnvironment(src);           // shares the SEAL objects
                dst.device = new CnDevice(src.parameters, src.relinKeys.DecompositionBitCount, src.galoisKeys.DecompositionBitCount, device,
                                          AtomicSealBfvEncryptedEnvironment.DeferredSubmission);
                // the keys AND the key-switch convention the source's start-up self-test settled on ("ks_xi") arrive together: cn_ctx_broadcast_keys copies both
                // (round 5, ADVICE r04 - before, a replica kept its own default and returned rc 0 and garbage when the source had flipped the convention)
                CnHip.Check(CnHip.cn_ctx_broadcast_keys(new IntPtr[] { src.device.Ctx, dst.device.Ctx }, 2));
                dst.SelfTestReport = src.SelfTestReport;
                return dst;
            }).ToArray();
            FreeComputationEnv(env);
            return res;
        }

        /// <summary>device-side operation counters of every plaintext-prime channel (OperationsCount of the reference, per context)</summary>
        public CnStats[] DeviceStatistics(bool reset = false)
        {
            var env = Reference;
            var res = env.Environments.Select(e => { CnHip.Check(CnHip.cn_stats_get(e.device.Ctx, out CnStats s, reset ? 1 : 0)); return s; }).ToArray();
            FreeComputationEnv(env);
            return res;
        }
    }
}
#endif
