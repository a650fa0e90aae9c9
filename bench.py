#!/usr/bin/env python
"""bench.py -- encrypted images/s of CryptoNets-MNIST (N=8192, 5 RNS limbs, 2 plaintext primes) on MI355X.

One step = one 8192-image ciphertext batch through the five evaluated layers (the reference's "Batch-Time"
window, CryptoNets/CryptoNets.cs:31,74) for BOTH plaintext-prime channels, inputs resident in HBM.
N GPUs: one process per GPU (torch.distributed, backend nccl = RCCL), every rank evaluates its own
independent batch (weak scaling, no data-path collective); evaluation keys are broadcast once from rank 0.
Prints ONE JSON line on rank 0 with `roofline` (the N=8192 NTT kernel, HIP-event timed), `key_switch` and `square` (the two kernel
families that are 90 % of the batch, priced against FP64 issue with instruction counts read from the built code object) and, at N=1,
`unchanged_caller` (the reference's per-ciphertext call pattern replayed on the C ABI), `lola` / `cifar` (BASELINE configs 4 and 5 as
child processes: `--workload lola|cifar`, each verified inside its run; `--no-single-image` skips them) and `cpu_baseline` (the CPU
oracle = port of the reference's SEAL path, timed on a bounded sample).  The whole default line takes about a minute.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def uniform_ct_words(rng, q, n, count, polys=2):
    """Uniform RNS residues: the distribution of fresh BFV ciphertext / key words."""
    out = np.empty((count, polys, len(q), n), dtype=np.uint64)
    for j, qj in enumerate(q):
        out[:, :, j, :] = rng.integers(0, qj, size=(count, polys, n), dtype=np.uint64)
    return out.reshape(count, -1)


def effective_cores():
    """(cores the process may use at once, visible logical CPUs, cgroup quota string): a container with a CPU quota (cgroup v2 cpu.max) below its
    visible CPU count gets quota / period CPUs' worth of time - the number a runtime's processor count reports (.NET's Environment.ProcessorCount
    honours it) and the only honest `cores` for an all-core CPU figure"""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
        q, per = quota.split()
        if q != "max":
            return max(1, min(visible, -(-int(q) // int(per)))), visible, quota
    except (OSError, ValueError):
        pass
    return visible, visible, quota


def _omp_threads(n):
    """OpenMP team size of the oracle library for the parallel regions that follow (omp_set_num_threads of the loaded runtime)"""
    import ctypes
    for name in ("libgomp.so.1", "libomp.so", "libiomp5.so"):
        try:
            ctypes.CDLL(name).omp_set_num_threads(int(n))
            return True
        except OSError:
            continue
    return False


def _cpu_sample(o, layers, p, ns, min_seconds=0.0):
    """seconds per 8192-image batch and plaintext prime, extrapolated from ns-sized samples of every layer (the layers' own weight rows).
    Each sample is repeated until it has run for `min_seconds` (all-core run: every core gets several items of every layer type and works
    for about a second per type, so that thread start-up and the tail of the dynamic schedule do not decide the figure)."""
    from cryptonets_amd import cryptonets_mnist as cm
    rng = np.random.default_rng(7)
    cts = uniform_ct_words(rng, o.q, o.n, 64)
    W = [cm.residues(L["W"], p) for L in layers]
    rows = lambda Wl, cnt: np.ascontiguousarray(Wl[np.arange(cnt) % Wl.shape[0]])

    def timed(fn, items):
        fn()                                              # warm: arenas, page tables, OpenMP team
        reps, t0 = 0, time.perf_counter()
        while True:
            fn()
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= min_seconds:
                return dt / (reps * items)
    big = np.tile(cts, (14, 1))[:845]
    wc, w3, w5 = rows(W[0], 4 * ns), rows(W[1], ns), rows(W[2], ns)
    t_conv = timed(lambda: o.scalar_gemm(cts[:25], wc), 4 * ns)
    t_d3 = timed(lambda: o.scalar_gemm(big, w3), ns)
    t_d5 = timed(lambda: o.scalar_gemm(big[:100], w5), ns)
    sq = np.tile(cts, (max(1, (2 * ns + 63) // 64), 1))[:2 * ns]
    t_sq = timed(lambda: o.mul_relin_batch(sq, sq), 2 * ns)
    return 845 * t_conv + 945 * t_sq + 100 * t_d3 + 10 * t_d5


def cpu_baseline(cores, layers):
    """Time the CPU oracle (port of the SEAL 3.2 path the reference runs) on a bounded sample of the same workload - the same weight
    rows the GPU run used - and extrapolate to one 8192-image batch over both plaintext primes: on all host cores (OpenMP over
    ciphertexts, mirroring ParallelProcessInEnv; large blocks kept in the per-thread malloc arenas - cno_tune_allocator - and every
    layer type sampled for about a second per core) and on ONE thread (SURVEY 8d asks for both).  `scaling_efficiency` = the all-core rate
    over cores x the single-thread rate."""
    from oracle.cno import Oracle, lib as oracle_lib
    from cryptonets_amd import cryptonets_mnist as cm
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    oracle_lib().cno_tune_allocator()
    p = cm.PLAIN_PRIMES[0]
    o = Oracle(cm.N, p, dbc=10, gdbc=20)
    o.keygen(1, galois=False)
    single = None
    if _omp_threads(1):
        t1 = 2 * _cpu_sample(o, layers, p, 2)
        single = {"value": round(8192.0 / t1, 2), "unit": "images/s", "cores": 1, "seconds_per_batch": round(t1, 1)}
        _omp_threads(cores)
    ns = max(8, 2 * cores)                            # several items per core and layer type
    total = 2 * _cpu_sample(o, layers, p, ns, min_seconds=1.0)
    out = {"value": round(8192.0 / total, 1), "unit": "images/s", "cores": cores, "kind": "port", "single_thread": single,
           "sample": "oracle (C restatement of SEAL 3.2 BFV, OpenMP over ciphertexts, malloc arenas tuned) timed on %d conv outputs, %d dense-845 outputs, "
                     "%d dense-100 outputs, %d square+relinearize ciphertexts of the N=8192 k=5 workload with the run's own weight rows, each sample repeated "
                     "for >= 1 s, extrapolated to 845/100/10/945 x 2 primes (%.1f s per batch); single_thread: the same on one thread from 8/2/2/4 items"
                     % (4 * ns, ns, ns, 2 * ns, total)}
    eff, visible, quota = effective_cores()
    out["host"] = {"visible_cpus": visible, "cgroup_cpu_max": quota, "effective_cores": eff}
    if single:
        out["scaling_efficiency"] = round(out["value"] / (cores * single["value"]), 3)
    # what the host grants to pure compute: the same register-only loop on 1 thread and on all of them
    try:
        import ctypes
        L = oracle_lib()
        L.cno_compute_probe.restype = ctypes.c_uint64
        L.cno_compute_probe.argtypes = [ctypes.c_uint64]
        _omp_threads(1)
        t0 = time.perf_counter(); L.cno_compute_probe(20_000_000); t_one = time.perf_counter() - t0
        _omp_threads(cores)
        L.cno_compute_probe(1_000_000)
        t0 = time.perf_counter(); L.cno_compute_probe(20_000_000); t_all = time.perf_counter() - t0
        out["host_compute_speedup"] = round(cores * t_one / t_all, 1)
        out["host_note"] = ("a register-only loop runs %.0f x faster on %d OpenMP threads than on one: the ceiling for any all-core figure on this host"
                            % (cores * t_one / t_all, cores))
    except Exception as ex:
        out["host_compute_speedup"] = None
        out["host_note"] = str(ex)[:120]
    return out


def single_image_workload(args, rank, world, local, dist, torch, result_fd):
    """--workload lola | cifar: the single-image networks of BASELINE configs 4 / 5 (LoLa-MNIST, LoLa-CIFAR shapes) on N GPUs.
    --shard images (default): every rank evaluates its own images with its own replica of all plaintext-prime channels - round-robin over
    ranks, no data-path collective (throughput; weak scaling).  --shard primes: the plaintext primes of ONE inference are dealt to the ranks
    (`distributed.shard_primes`), every rank evaluates the whole network for its primes and the only exchange is the client-side CRT join
    of the decrypted residues (latency of one image; strong scaling, at most #primes ranks do work).  Each prime channel has its own keys,
    generated where the channel lives: no key broadcast is needed for either split in this benchmark (a server that replicates a
    client's channel on several GPUs receives that client's evaluation keys once - cn_ctx_broadcast_keys / the C3 path above)."""
    from cryptonets_amd import cryptonets_mnist as cm, networks
    from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
    from cryptonets_amd.distributed import crt_join_over_ranks, max_over_ranks, shard_primes
    dev = torch.device("cuda", local)
    cifar = args.workload == "cifar"
    name = "LoLaCifar" if cifar else "LoLa"
    parms = dict(networks.FACTORY_PARAMETERS[name], device=local)
    all_primes = list(parms["primes"])
    if args.shard == "primes":
        parms["primes"] = shard_primes(all_primes, rank, world)
    active = len(parms["primes"]) > 0
    exchanges = []
    if args.client_seed is not None:
        parms["client_seed"] = args.client_seed
    if args.shared_keys:
        if args.shard != "images":
            raise SystemExit("--shared-keys goes with --shard images (with --shard primes every rank owns different plaintext primes: nothing to share)")
        from cryptonets_amd.client import SharedKeyDeviceClient
        # with_client_keys: every benchmark rank also plays the data owner (it encrypts its own images and decrypts its logits to verify them), so the client's
        # public and secret key travel too - the library's default is evaluation keys only (distributed.BroadcastKeys)
        parms["device_client_factory"] = lambda ctx, t: SharedKeyDeviceClient(ctx, dist, dev, 0, seed=None if args.client_seed is None else args.client_seed ^ t,
                                                                                exchanges=exchanges, with_client_keys=True)
    rng = np.random.default_rng(5)
    if cifar:
        qz = lambda a, sc: np.rint(a * sc) / sc
        W = [qz(rng.normal(0, 0.05, 83 * 192), 256), qz(rng.normal(0, 0.02, 112 * 8300), 512), qz(rng.normal(0, 0.05, 10 * 5488), 512)]
        B = [qz(rng.normal(0, 0.05, 83), 256), qz(rng.normal(0, 0.05, 112), 512), qz(rng.normal(0, 0.05, 10), 512)]
        images = rng.integers(0, 256, size=(args.warmup + args.steps, 3 * 32 * 32)).astype(float)
    else:
        w = dict(zip(("Weights_0", "Weights_1", "Biases_2", "Weights_3", "Biases_3"), cm.reference_weights()))
        images = cm.synthetic_images(args.warmup + args.steps, seed=1000 + (rank if args.shard == "images" else 0))
    results, dt, verified, kept = [], 0.0, None, None
    if active:
        Factory = EncryptedSealBfvFactory(**parms)
        env = Factory.AllocateComputationEnv()
        if cifar:
            reader = networks.cifar_reader(Factory=Factory)
            net = networks.LoLaCifar(Factory, reader, W, B, timing=False)
        else:
            tsv = "/tmp/bench_lola_rank%d.tsv" % rank
            with open(tsv, "w") as f:
                for img in images:
                    f.write("7\t784\t" + "\t".join("%d:%d" % (i, int(img[i])) for i in np.nonzero(img)[0]) + "\n")
            reader = networks.lola_reader(name, tsv, Factory=Factory)
            net = networks.LOLA_NETWORKS[name](Factory, reader, w)
        net.PrepareNetwork()
        layers = list(networks._chain(net))[::-1]            # reader, encrypt, evaluated layers ...

        def sync():
            for e in env.Environments:
                e.ctx.sync()
    # EncryptLayer runs outside the timed window, like the reference's "Prediction-Time" / "Inference-Time" brackets (LoLaCryptonets.cs:64-115,
    # LolaCifarCryptoNet.cs:57,128): every image of the run is encrypted first
    encrypted = []
    for it in range(args.warmup + args.steps if active else 0):
        if cifar:
            reader.Features = images[it] / 256.0
            encrypted.append(layers[1].GetNext())
        else:
            encrypted.append(layers[1].Apply(layers[0].GetNext()))
    if active:
        sync()
    samples = []                                               # wall time of every timed image of this rank (the contexts are synchronised after each)
    for it in range(args.warmup + args.steps):
        if it == args.warmup:
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
        if not active:
            continue
        t_img = time.perf_counter()
        m = encrypted[it]
        for li, L in enumerate(layers[2:]):
            m2 = L.Apply(m)
            if m2 is not m:
                if m is kept:
                    pass                                  # (the big dense layer's output of the last image: decrypted after the timed window)
                else:
                    m.Dispose()
            m = m2
            if cifar and li == 3 and it == args.warmup + args.steps - 1:
                kept = m                                  # layers[5] = LLDenseLayer 5488 x 16268
        sync()
        if it >= args.warmup:
            results.append(m)
            samples.append(time.perf_counter() - t_img)
        else:
            m.Dispose()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, dev, dist)
    # ---- outside the timed window: decrypt, CRT-join (over ranks when the primes are split), compare with the exact integer model
    if not cifar:
        M = 1
        for p_ in all_primes:
            M *= p_
        verified = True
        for i, m in enumerate(results if active else [None] * args.steps):
            res = {}
            if active:
                col = m.GetColumn(0)
                res = {p_: np.asarray(a._decrypt_ints(e), dtype=object)[:10] for p_, a, e in zip(parms["primes"], col.eVectors, env.Environments)}
            if args.shard == "primes":
                joined = crt_join_over_ranks(res, all_primes, dist)
                want = cm.centred(cm.int_logits(w, cm.synthetic_images(args.warmup + args.steps, seed=1000)[args.warmup + i]), M)
            else:
                joined = crt_join_over_ranks(res, all_primes, None)
                want = cm.centred(cm.int_logits(w, images[args.warmup + i]), M)
            verified = verified and [int(x) for x in joined] == [int(x) for x in want]
        if dist is not None:
            flag = torch.tensor([1 if verified else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            verified = bool(flag.item())
    if cifar and active and kept is not None:
        # LoLa-CIFAR with the reference's 8 limbs keeps a positive noise budget through the 5488 x 16268 dense layer but not through the
        # whole network (DESIGN: a property of the reference's operation sequence): the measured run is verified THERE - all 5488 outputs of
        # the last timed image, every plaintext prime, against the exact integer model
        verified = True
        img = images[args.warmup + args.steps - 1]
        for p_, a, e in zip(parms["primes"], kept.GetColumn(0).eVectors, env.Environments):
            got = [int(v) for v in a._decrypt_ints(e)]
            want = [int(v) for v in networks.lola_cifar_dense_model(reader, layers[2], W, B, img, p_)]
            verified = verified and got == want
        kept.Dispose()
        if dist is not None:
            flag = torch.tensor([1 if verified else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            verified = bool(flag.item())
    import hashlib
    digest = hashlib.sha256()                                  # the result ciphertext WORDS of this rank's timed images (reproducible under --client-seed)
    for m in results:
        for a, e in zip(m.GetColumn(0).eVectors, env.Environments):
            digest.update(e.ctx.ct_download(a.encData.h, a.encData.first, a.encData.count).tobytes())
    for m in results:
        m.Dispose()
    if rank == 0:
        images_done = args.steps * (world if args.shard == "images" else 1)
        out = {"metric": "encrypted images/sec (%s single-image inference, N=%d)" % ("LoLa-CIFAR shapes" if cifar else "LoLa-MNIST", parms["n"]),
               "value": round(images_done / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak" if args.shard == "images" else "strong",
               "vs_baseline": None, "dtype": "u64", "data": "synthetic images" + (", synthetic weights of the reference's shapes" if cifar else ", the reference's trained weights"),
               "ms_per_image": {"min": round(1e3 * min(samples), 2), "median": round(1e3 * float(np.median(samples)), 2), "all": [round(1e3 * x, 2) for x in samples]} if samples else None,
               "verified_against_integer_model": verified, "result_words_sha256": digest.hexdigest(),
               "launcher": os.environ.get("BENCH_LAUNCHER", "external" if "WORLD_SIZE" in os.environ else "none"), "process_group": "nccl" if dist is not None else None,
               "keys": ("one client: rank 0 ran KeyGenerator, every evaluation key broadcast with RCCL and adopted in place; the client keys travel too (explicit opt-in: every benchmark rank decrypts its own logits)" if args.shared_keys else "every rank is its own client (own KeyGenerator)"),
               "key_broadcast": ({"bytes": sum(b for b, _ in exchanges), "ms": round(1e3 * sum(t for _, t in exchanges), 2), "contexts": len(exchanges),
                                  "GB_per_s": round(sum(b for b, _ in exchanges) / max(1e-9, sum(t for _, t in exchanges)) / 1e9, 2),
                                  "keys_per_context": 1 + 2 * (parms["n"].bit_length() - 2) + 2,
                                  "note": "outside the timed window; relinearisation key + the default Galois key set + public and secret key per plaintext prime; "
                                          "host staging of the root's words included (cn_get_key converts the FP64 key image back to u64 words)"} if args.shared_keys else None),
               "verified_what": ("all 5488 outputs of the 5488 x 16268 dense layer of the last timed image, every plaintext prime (the 8-limb network has no "
                                 "noise budget left behind its second squaring - a property of the reference's operation sequence)") if cifar
                                else "the 10 logits of every timed image, CRT-joined over the plaintext primes",
               "config": {"workload": "%s (BASELINE config %d), one image per step and %s" % (name, 5 if cifar else 4, "rank" if args.shard == "images" else "job"),
                          "plaintext_primes": all_primes, "sharding": args.shard,
                          "parallelism": ("images round-robin over %d ranks, full replica per rank" % world) if args.shard == "images"
                                         else ("plaintext primes dealt to %d ranks, CRT join of decrypted residues on the client" % world)}}
        if cifar and active and world == 1:
            out["key_switch"] = cifar_key_switch_block(env.Environments[0].ctx, 1e3 * dt / args.steps, len(parms["primes"]))
        if not cifar and world == 1 and not args.no_unchanged_caller:
            # the UNCHANGED per-call sequence of the reference's LL layers (one vector method per row / column / map), recorded at the C ABI and
            # replayed from C++ (tools/lola_unchanged_caller.py): ms per image next to the batched conveniences measured the same way
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import lola_unchanged_caller
                for e in env.Environments:                     # the measured contexts give their hardware queues back (a process has four: cn_api.hip pick_stream)
                    e.ctx.close()
                rows = lola_unchanged_caller.measure(name, reps=max(5, min(20, args.steps)), device=local)
                sel = lambda lit, dfr, host: [r for r in rows if r["pattern"].startswith("unchanged") == lit and r["pattern"].endswith("deferred submission") == dfr and r["host"].startswith(host)][0]
                # host pattern of the reference: ForEveryEncryptedVector runs one Task per plaintext prime and joins them after every vector call
                # (EncryptedSealBfvVector.cs:225-236) = the rows "one thread per plaintext prime, joined after every call"
                host = "C++ replay, one thread per plaintext prime, joined"
                lit, imm, bat = sel(True, True, host), sel(True, False, host), sel(False, False, host)
                out["unchanged_caller"] = {"ms_per_image": lit["ms_per_image"], "logits_exact": lit["logits_exact"], "calls_per_prime": lit["calls_per_prime"],
                                           "launches_per_prime": lit.get("launches_per_prime"), "every_call_launched_on_its_own_ms": imm["ms_per_image"],
                                           "batched_from_the_same_host_ms": bat["ms_per_image"], "batched_calls_per_prime": bat["calls_per_prime"],
                                           "frac_of_batched": round(bat["ms_per_image"] / lit["ms_per_image"], 3),
                                           "pattern": "the per-call sequence of the reference's unchanged LL layers (one vector method per row / column / map), recorded at the C "
                                                      "ABI, replayed from C++ (one thread per plaintext prime, joined after every call) with cn_set_option(defer, 1)", "all_rows": rows}
            except Exception as ex:
                out["unchanged_caller"] = {"error": str(ex)[:300]}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cifar_key_switch_block(g, ms_per_image, primes):
    """The kernel that IS a LoLa-CIFAR image: the N = 16384, k = 8 batch key switch - one launch of k_keyswitch_pair14 per
    rotate-and-add link of the 5488-row SumAllSlots chain (14 links per plaintext prime).  Timed here with HIP events on the context's stream over a
    5488-ciphertext array of random residues, and priced against the FP64 issue rate measured in this process right behind it (cn_valu_issue_time).
    Instructions per thread: DYNAMIC counts of the built kernel from the committed counter passes (profiles/r05_ks14_counters.json: SQ_INSTS_VALU* per
    launch / waves per launch) - rocprofv3 wraps a command, so this process cannot collect them itself; the file and the launch geometry it was collected on
    are named.  Bound: VALU issue (FP64 and the other vector instructions share one issue port per SIMD); HBM-level bytes from the same passes."""
    rows = 5488
    n, k = g.n, g.k
    rng = np.random.default_rng(7)
    one = np.concatenate([rng.integers(0, int(q), size=n, dtype=np.uint64) for _ in range(2) for q in g.q])
    blk = np.repeat(one[None, :], 64, axis=0)
    h = g.ct_alloc(rows)
    for i in range(0, rows, 64):
        g.ct_upload(h, i, blk[:min(64, rows - i)])
    g.sum_slots(h, 0, rows, 0); g.sync()
    g.time_begin()
    g.sum_slots(h, 0, rows, 0)
    chain_ms = g.time_end()
    fp64_ns = g.fp64_issue_ns(iters=4096, launches=6)
    valu_ns = g.valu32_issue_ns(iters=16384, launches=6)
    g.free(h)
    links = 1 + (n // 2).bit_length() - 1                       # the column swap + log2(N/2) row rotations
    link_ms = chain_ms / links
    blk_ = {"kernel": "k_keyswitch_pair14 (%d ciphertexts x %d output limbs: 2 x %d digit transforms + 4 inverse half-transforms per workgroup, one launch per rotate-and-add link)" % (rows, k, k),
            "bound": "valu issue (fp64)", "ms_per_link": round(link_ms, 3), "links_per_prime": links, "ms_per_chain": round(chain_ms, 2),
            "share_of_image": round(primes * chain_ms / ms_per_image, 3),
            "share_note": "this chain alone on one stream, timed behind the images (the clock is at its lowest there); inside an image the chains of the plaintext primes run on two hardware queues and fill each other's last partial wave of workgroups"}
    try:
        import glob
        src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ks14_counters.json")))[-1]
        c = json.load(open(src))
        waves = rows * k * 8
        fp64 = c["fp64_per_wave"]; other = c["valu_per_wave"] - fp64
        floor_fp64 = waves * fp64 / 1024 * fp64_ns * 1e-6
        floor_valu = waves * (fp64 * fp64_ns + other * valu_ns) / 1024 * 1e-6
        blk_.update({"fp64_per_thread": round(fp64), "valu_other_per_thread": round(other), "counts_source": os.path.relpath(src, ROOT) + " (committed rocprofv3 --pmc passes of the same launch geometry; not measured by this run)",
                     "fp64_ns_per_instr_in_situ": round(fp64_ns, 3), "valu32_ns_per_instr_in_situ": round(valu_ns, 3),
                     "fp64_issue_floor_in_situ_ms": round(floor_fp64, 2), "frac": round(floor_fp64 / link_ms, 3),
                     "valu_issue_floor_in_situ_ms": round(floor_valu, 2), "frac_valu_in_situ": round(floor_valu / link_ms, 3),
                     "hbm_level_bytes_per_link": c.get("hbm_bytes_per_launch"), "hbm_frac": round(c["hbm_bytes_per_launch"] / (link_ms * 1e-3) / 8e12, 3) if c.get("hbm_bytes_per_launch") else None,
                     "algorithmic_bytes_per_link": rows * 2 * k * n * 8 * 2,          # x + rot(x) in place: every ciphertext read once and written once (the Galois key, 16 MiB, is shared by all)
                     "rounds_1_to_4_ms_per_link": c.get("rounds_1_to_4_ms_per_link")})
    except Exception as ex:
        blk_.update({"frac": None, "counts_error": str(ex)[:200]})
    return blk_


def single_image_lines(gpu_index):
    """BASELINE configs 4 and 5 on the driver's line: `python bench.py --workload lola|cifar` run as child processes of the default
    workload (a process of their own: every plaintext-prime channel then gets a hardware queue of its own, cn_api.hip pick_stream) on the same GPU,
    bounded - LoLa-MNIST 20 timed images + the unchanged-caller replay, LoLa-CIFAR 2 warm-up + 3 timed images (min / median / all samples travel with the line) - each verified against the exact integer
    model inside its run.  A child that fails or exceeds its time limit is reported with the reason instead of a number."""
    import subprocess
    # (OMP_*: cpu_baseline sets OMP_PROC_BIND=spread for the oracle's OpenMP team in THIS process; inherited, it binds the child's initial thread to one core
    # and every std::thread of the unchanged-caller replay with it - the LoLa child then crawls: 240 s instead of 20, profiles/r04_notes)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BENCH_LAUNCHER",
                                                            "BENCH_FORCE_DIST", "BENCH_SELF_LAUNCH", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")
           and not k.startswith("OMP_")}
    env["HIP_VISIBLE_DEVICES"] = os.environ.get("HIP_VISIBLE_DEVICES", "").split(",")[gpu_index] if os.environ.get("HIP_VISIBLE_DEVICES") else str(gpu_index)

    def child(workload, steps, warmup, limit):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup)]
        t0 = time.perf_counter()
        proc = subprocess.Popen(cmd, env=dict(env, PYTHONFAULTHANDLER="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        try:
            so, se = proc.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            import signal
            proc.send_signal(signal.SIGABRT)                   # faulthandler: the Python stacks of all threads go to stderr - the reason travels with the line
            try:
                so, se = proc.communicate(timeout=10)
            except subprocess.TimeoutExpired:
                proc.kill()
                so, se = proc.communicate()
            return None, {"skipped": "child exceeded its %d s limit" % limit, "command": " ".join(cmd[1:]), "where": se.decode(errors="replace")[-1500:]}
        wall = time.perf_counter() - t0
        lines = [l for l in so.decode(errors="replace").splitlines() if l.startswith("{")]
        if proc.returncode or not lines:
            return None, {"error": "rc %d: %s" % (proc.returncode, se.decode(errors="replace")[-300:]), "command": " ".join(cmd[1:])}
        d = json.loads(lines[-1])
        return d, {"command": "python bench.py " + " ".join(cmd[2:]), "child_wall_s": round(wall, 1)}

    d, lola = child("lola", 20, 3, int(os.environ.get("BENCH_CHILD_LIMIT", "240")))
    if d is not None:
        u = d.get("unchanged_caller") or {}
        lola.update({"metric": d["metric"], "ms_per_image": d["ms_per_step"], "images_per_s": d["value"], "steps": d["steps"],
                     "ms_per_image_samples": d.get("ms_per_image"),
                     "verified": d["verified_against_integer_model"], "verified_what": d["verified_what"], "plaintext_primes": d["config"]["plaintext_primes"],
                     "unchanged_caller_ms": u.get("ms_per_image"), "unchanged_caller_logits_exact": u.get("logits_exact"), "launches_per_prime": u.get("launches_per_prime"),
                     "calls_per_prime": u.get("calls_per_prime"), "batched_from_the_same_host_ms": u.get("batched_from_the_same_host_ms"),
                     "unchanged_frac_of_batched": u.get("frac_of_batched"), "unchanged_caller_error": u.get("error")})
    d, cifar = child("cifar", 3, 2, 300)                       # steady state: two untimed images (work arrays at their final size), three timed
    if d is not None:
        cifar.update({"metric": d["metric"], "s_per_image": round(d["ms_per_step"] / 1e3, 3), "ms_per_image": d.get("ms_per_image"), "steps": d["steps"], "warmup": d["warmup"],
                      "verified": d["verified_against_integer_model"], "verified_what": d["verified_what"], "plaintext_primes": d["config"]["plaintext_primes"], "data": d["data"],
                      "key_switch": d.get("key_switch")})
    return lola, cifar


def _conv_tile():
    """BENCH_CONV_TILE=2 or 1x2: outputs of neighbouring positions share a gather list (same outputs; developer A/B switch)."""
    v = os.environ.get("BENCH_CONV_TILE", "1")
    return tuple(int(x) for x in v.split("x")) if "x" in v else int(v)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(gpus, argv):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves - the same
    command line the driver uses for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py ...`), one process per GPU, LOCAL_RANK -> device.  Rank 0 prints the ONE JSON line on the inherited stdout;
    the exit status is torchrun's."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def stub_workload(args, rank, world, result_fd):
    """--stub: the launcher / rendezvous / barrier / max-over-ranks / one-line plumbing of the N-rank run with NO device work (gloo on
    CPU; tests/test_bench_launch.py runs `bench.py --gpus 2 --stub` here, where there is no GPU).  The line is marked `"stub": true` and
    carries no throughput claim: `value` counts the stub's sleep steps."""
    import torch
    import torch.distributed as dist
    from cryptonets_amd.distributed import broadcast_words, max_over_ranks, shard_batches
    d = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        d = dist
    words = np.arange(1024, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) if rank == 0 else None
    got = broadcast_words(words, 1024, 0, "cpu", d).numpy().view(np.uint64)          # the key broadcast of the real run, on gloo
    ok = bool(np.array_equal(got, np.arange(1024, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
    mine = shard_batches(args.steps * world, rank, world)                               # every rank its own batches
    for _ in range(args.warmup):
        time.sleep(0.01)
    if d is not None:
        d.barrier()
    t0 = time.perf_counter()
    for _ in mine:
        time.sleep(0.01 * (1 + rank))                                                  # rank r is slower: the line must carry the MAX
    if d is not None:
        d.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, "cpu", d)
    cover = sorted(mine) == list(range(args.steps)) if world == 1 else None
    if d is not None:
        flag = torch.tensor([1 if ok and len(mine) == args.steps else 0], dtype=torch.int32)
        d.all_reduce(flag, op=d.ReduceOp.MIN)
        ok = bool(flag.item())
        shards = [None] * world                                                        # shard cover: every batch of the job on exactly one rank
        d.all_gather_object(shards, [int(b) for b in mine])
        cover = sorted(b for sh in shards for b in sh) == list(range(args.steps * world))
    # what the real run does beside the timed region, and on which ranks: the CPU baseline, the single-image children and the unchanged caller are rank-0,
    # world-1 work (main(): `rank == 0 and world == 1`) - at N > 1 no rank runs them, so the driver's SCALE lines time the sharded batches only
    side = {"cpu_baseline": rank == 0 and world == 1, "single_image_children": rank == 0 and world == 1, "unchanged_caller": rank == 0 and world == 1}
    if rank == 0:
        out = {"metric": "encrypted images/sec (CryptoNets-MNIST, N=8192)", "stub": True, "value": round(8192 * args.steps * world / dt, 1), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "none (launcher stub)", "plumbing_ok": ok, "shard_cover_ok": bool(cover),
               "rank0_side_work": side,
               "config": {"workload": "STUB: no device work - launcher, gloo rendezvous, key broadcast, barrier and MAX-over-ranks timing only"}}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if d is not None:
        d.barrier()
        d.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--stub", action="store_true", help="launcher plumbing only (gloo, no device work): what the CPU test of `--gpus N` runs")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weights", choices=("trained", "synthetic"), default="trained",
                    help="trained: the reference's CryptoNets/Weights.cs (shipped as package data); synthetic: random-init weights of the same shapes")
    ap.add_argument("--no-relinearize-late", action="store_true", help="skip the opt-in program variant with Relinearize behind the dense layers")
    ap.add_argument("--no-unchanged-caller", action="store_true", help="skip the per-ciphertext-call replay of the reference's unchanged layers")
    ap.add_argument("--caller-threads", type=int, default=0, help="threads of the unchanged-caller replay; 0 = the processor count a runtime reports here "
                    "(cgroup CPU quota honoured) - the reference's Defaults.ThreadCount = Environment.ProcessorCount")
    ap.add_argument("--serialize", action="store_true", help="sync after every plaintext-prime channel (clean per-kernel profiles)")
    ap.add_argument("--stagger", type=int, default=int(os.environ.get("BENCH_STAGGER", "0")),
                    help="1: the plaintext-prime channels run half a batch apart (key switch of one beside the HBM-bound layers of the other; the default of rounds 3-6); "
                         "0 (default): the plain loop, prime after prime - cn_mul_relin pipelines each prime's squarings in parts by itself")
    ap.add_argument("--workload", choices=("cryptonets", "lola", "cifar"), default="cryptonets",
                    help="cryptonets: BASELINE config 3, the headline metric (default); lola / cifar: the single-image networks of configs 4 / 5")
    ap.add_argument("--shard", choices=("images", "primes"), default="images", help="lola / cifar on N GPUs: independent images per rank, or the plaintext primes of one image")
    ap.add_argument("--shared-keys", action="store_true", help="lola / cifar, --shard images: ONE client's keys on every rank - rank 0 runs KeyGenerator, the relinearisation key, "
                    "every Galois key and the client keys are broadcast with RCCL and adopted in place (default: every rank is its own client)")
    ap.add_argument("--client-seed", type=int, default=None, help="lola / cifar: reproducible keys and encryption randomness (tests: two runs must give the same ciphertext words)")
    ap.add_argument("--no-single-image", action="store_true", help="default workload: skip the LoLa-MNIST / LoLa-CIFAR sub-benchmarks (BASELINE configs 4 and 5) of the line")
    args = ap.parse_args()

    # N > 1 without a launcher around us: start the ranks ourselves (the driver's own torchrun line sets WORLD_SIZE and is honoured as is)
    if (args.gpus > 1 or os.environ.get("BENCH_SELF_LAUNCH")) and "WORLD_SIZE" not in os.environ:      # BENCH_SELF_LAUNCH=1: also at N = 1 (GPU test of the launcher)
        os.environ["BENCH_LAUNCHER"] = "self (torch.distributed.run)"
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print("bench.py: --gpus %d but the launcher started %s ranks: the launcher wins" % (args.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)

    # the contract is ONE JSON line on stdout: RCCL / HIP runtime banners are C stdio writes to fd 1 (flushed at exit), so fd 1 is
    # pointed at stderr for the whole run and the result line goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.stub:
        return stub_workload(args, rank, world, result_fd)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libcnhip has no CPU path")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") or os.environ.get("BENCH_LAUNCHER"):      # BENCH_FORCE_DIST=1: exercise the RCCL plumbing with a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    if args.workload != "cryptonets":
        return single_image_workload(args, rank, world, local, dist, torch, result_fd)

    from cryptonets_amd._native import Context
    from cryptonets_amd import cryptonets_mnist as cm
    from cryptonets_amd.distributed import broadcast_words, max_over_ranks

    weights = cm.reference_weights() if args.weights == "trained" else cm.synthetic_weights(1)
    layers = cm.layer_tables(*weights, conv_tile=_conv_tile())      # gather-list tiling: same outputs
    images = cm.synthetic_images(cm.N, seed=1000 + rank)            # this rank's batch (independent batches per GPU)
    x_int = np.rint(images * cm.NORMALIZATION * cm.INPUT_SCALE).astype(np.int64)
    dev = torch.device("cuda", local)
    chans, key_tensors = [], []
    for p in cm.PLAIN_PRIMES:
        g = Context(cm.N, p, dbc=10, gdbc=20, device=local)
        # Keys: rank 0 runs KeyGenerator on its GPU; the public evaluation (relinearisation) key is broadcast with RCCL over
        # xGMI and adopted in place by every rank.  (In this benchmark every rank also plays the data owner, so the public
        # and secret keys travel too; a real server only ever receives the evaluation keys.)
        if rank == 0:
            g.keygen(0xC0FFEE ^ p, galois=False)
        parts = []
        for which, words in ((0, g.key_words(False)), (2, g.ctw), (3, g.ctw // 2)):
            kw = g.get_key(which) if rank == 0 else None
            parts.append(broadcast_words(kw, words, 0, dev, dist))
        torch.cuda.synchronize()
        if rank != 0:
            g.set_public_key(parts[1].cpu().numpy().view(np.uint64))
            g.set_secret_key(parts[2].cpu().numpy().view(np.uint64))
        if rank != 0 or dist is not None:
            g.set_relin_key_device(parts[0].data_ptr(), parts[0].numel())       # adopt the broadcast buffer (converted in place)
        key_tensors.append(parts)
        ch = cm.CryptoNetsChannel(g, layers, cm.constant_plaintext(cm.N))
        # EncryptLayer on the device: 784 pixel columns -> BatchEncoder.Encode -> Encryptor.Encrypt (outside the timed window,
        # like the reference's TimingLayer placement)
        ph = g.pt_alloc(784)
        g.encode_batch(np.ascontiguousarray(np.mod(x_int, p).astype(np.uint64).T), ph, 0)
        g.encrypt(ph, 0, ch.h_in, 0, 784, seed=0xFEED ^ rank)
        g.free(ph)
        chans.append(ch)

    def step_staggered(split=1):
        # The two plaintext-prime channels are independent kernel chains on two streams.  Staggered by half a batch (device-side ordering between the contexts,
        # cn_ctx_wait_for; no host wait) the long FP64-bound key switch of one channel runs beside the HBM-bound layers of the other.  The default program of rounds 3-6 until
        # cn_mul_relin pipelined a batch in parts by itself ("sq_halves"): since then the PLAIN loop below is as fast or faster (profiles/r06_mulrelin_parts.txt).
        for i, ch in enumerate(chans):
            ch.front() if split == 1 else ch.front2()             # (2: experiment - the split behind the whole squaring layer)
            if not os.environ.get("BENCH_STAGGER_NOWAIT"):        # (A/B: the two halves without the device-side ordering)
                chans[(i + 1) % len(chans)].g.wait_for(ch.g)      # the next channel('s next batch) starts where this one's key switch starts
            ch.back() if split == 1 else ch.back2()

    def step():
        if args.stagger and not args.serialize and len(chans) > 1:
            return step_staggered(args.stagger)
        # the reference's own sequence, prime after prime (EncryptedSealBfvVector.cs:225-236 runs them as parallel tasks): five batched calls per prime; cn_mul_relin
        # pipelines the 845 squarings in three parts over two streams of its context, so a prime's key switches run beside its own and the other prime's Multiply steps
        for ch in chans:
            ch.forward()
            if args.serialize:
                ch.g.sync()

    def sync_all():
        for ch in chans:
            ch.g.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, torch.device("cuda", local), dist)

    # ---- the measured run is checked: decrypt the logits on the device and compare ALL 8192 slots x 10 outputs per prime with the
    # exact integer model of the network (same weights, same inputs); the digest of the final ciphertext words identifies the run
    # (fixed key / encryption seeds: a run with the relinearisation key adopted from a broadcast buffer must give the same digest)
    import hashlib
    verified = True
    digest = hashlib.sha256()
    for ch in chans:
        gg = ch.g
        dh = gg.pt_alloc(10)
        gg.decrypt(ch.h5, 0, 10, dh, 0)
        got = np.ascontiguousarray(gg.decode_batch(dh, 0, 10).T)
        gg.free(dh)
        verified = verified and bool(np.array_equal(got, cm.model_mod_p_dense(x_int, layers, gg.t)))
        digest.update(gg.ct_download(ch.h5, 0, 10).tobytes())

    if dist is not None:                                  # every rank checked its own batch: report the conjunction
        flag = torch.tensor([1 if verified else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item())

    # ---- roofline of the dominant kernel: the batched N=8192 RNS NTT, timed with HIP events on the ctx stream
    g = chans[0].g
    limbs = 845 * 2 * g.k
    ptr, _ = g.device_ptr(chans[0].h2)
    g.ntt_time(ptr, limbs, 0, False, 2)
    ms = g.ntt_time(ptr, limbs, 0, False, 20)
    alg_bytes = limbs * g.n * 8 * 2                      # each limb read once + written once (128 KiB per N=8192 limb)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    # HBM bytes per launch: PMC counters cannot be read from inside this process (rocprofv3 wraps the command), so the figure is the one
    # of the latest COMMITTED counter passes of this launch geometry (tools/visit_*.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs) and
    # is labelled with the file it comes from; null when no such file travels with the tree
    traffic, traffic_source = None, None
    try:
        import glob
        src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ntt_hbm_traffic.json")))[-1]
        prof = json.load(open(src))
        traffic = [v["hbm_traffic_corrected_bytes"] for kname, v in prof.items() if "k_ntt" in kname and "forward" in kname][0]
        traffic_source = os.path.relpath(src, ROOT) + " (committed rocprofv3 --pmc passes of the same launch; not measured by this run)"
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "k_ntt_rr (forward, %d limbs of N=%d u64)" % (limbs, g.n), "achieved": round(achieved, 1),
                "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_source,
                "ms_per_launch": round(ms, 4), "bytes_per_launch": alg_bytes}

    # ---- the kernel that takes the largest share of the batch (a third): the fused key switch of the first squaring layer, timed the same
    # way.  It is bound by FP64 issue, not by HBM: the ISA of the digit loop holds 1104 FP64 instructions per thread and digit (transform
    # + 2 x 16 multiply-accumulates), the two inverse transforms 1064 each, and the pipe issues one FP64 wave-instruction per 2.15 ns per
    # SIMD whatever the occupancy (profiles/r01_ubench_mulmod.txt); tools/ubench_ks.hip measures the arithmetic alone at the same 2.1 ms.
    key_switch = None
    if rank == 0:
        cts = 845
        t3, t2 = g.ct_alloc(cts, 3), g.ct_alloc(cts)
        g.multiply(chans[0].h1, 0, chans[0].h1, 0, t3, 0, cts)           # the products the layer relinearises
        g.relinearize(t3, 0, t2, 0, cts); g.sync()
        g.time_begin()
        for _ in range(5):
            g.relinearize(t3, 0, t2, 0, cts)
        ks_ms = g.time_end() / 5
        fp64_ns = g.fp64_issue_ns(iters=4096, launches=6)                 # ~3 ms of pure FP64 chains right behind the key switches: the issue rate NOW
        valu_ns = g.valu32_issue_ns(iters=16384, launches=6)              # and of full-rate 32-bit VALU instructions
        g.free(t3); g.free(t2)
        per_limb = [-(-int(q).bit_length() // 10) for q in g.q]           # base-2^10 digits of every source limb
        digits = sum(per_limb)
        key_switch = {"kernel": "k_keyswitch_rr (845 ciphertexts x %d output limbs, %d digit transforms each)" % (g.k, digits),
                      "share_of_batch": round(2 * ks_ms / (1e3 * dt / args.steps), 3), "bound": "fp64 issue", "ms_per_launch": round(ks_ms, 3),
                      "ns_per_limb_transform": round(ks_ms * 1e6 / (cts * g.k * (digits + 2)), 1)}
        try:        # FP64 instructions per thread from the disassembly of the BUILT kernel (tools/ks_isa_counts.py), priced at the measured
                    # issue rate of one FP64 wave-instruction per 2.15 ns per SIMD (profiles/r01_ubench_mulmod.txt)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import ks_isa_counts
            fp64_per_thread, isa = ks_isa_counts.fp64_per_thread(g.k, per_limb)
            floor_ms = cts * g.k * 8 * fp64_per_thread / 1024 * 2.15e-6   # 8 waves per workgroup, 4 SIMDs x 256 CUs
            floor_now = cts * g.k * 8 * fp64_per_thread / 1024 * fp64_ns * 1e-6
            key_switch.update({"fp64_per_thread": fp64_per_thread, "fp64_digit_loop": isa["fp64_digit_loop"], "fp64_tail_loop": isa["fp64_tail_loop"],
                               "fp64_issue_floor_ms": round(floor_ms, 3), "frac": round(floor_ms / ks_ms, 3),
                               "fp64_ns_per_instr_in_situ": round(fp64_ns, 3), "fp64_issue_floor_in_situ_ms": round(floor_now, 3),
                               "frac_in_situ": round(floor_now / ks_ms, 3)})
            # the SIMD's VALU port issues the FP64 AND the other vector instructions of the kernel (253 per digit: digit extraction,
            # selects, addressing), one at a time: the issue floor of the whole VALU stream, every non-FP64 instruction priced as a
            # full-rate one (a lower bound: 64-bit shifts and 32-bit multiplies take longer)
            valu_per_thread = ks_isa_counts.valu_per_thread(per_limb, isa)
            floor_valu = cts * g.k * 8 * (fp64_per_thread * fp64_ns + valu_per_thread * valu_ns) / 1024 * 1e-6
            key_switch.update({"valu_other_per_thread": valu_per_thread, "valu32_ns_per_instr_in_situ": round(valu_ns, 3),
                               "valu_issue_floor_in_situ_ms": round(floor_valu, 3), "frac_valu_in_situ": round(floor_valu / ks_ms, 3)})
        except Exception as ex:                                            # no disassembler / unrecognised code shape: no floor rather than a stale one
            key_switch.update({"fp64_issue_floor_ms": None, "frac": None, "isa_error": str(ex)[:200]})

    # ---- the second-largest family: the BEHZ product of the first squaring layer - k_behz_extend (Bsk limbs of the operand),
    # k_square_fused on the q base and on the Bsk base (2 forward + 3 inverse transforms + the tensor per (ciphertext, limb), operand parked in LDS),
    # k_behz_floor - timed as ONE chain with HIP events on the ctx stream (Evaluator.Multiply(a, a) of 845 ciphertexts) and priced against both
    # bounds: FP64 issue (ISA count of the BUILT k_square_fused from tools/ks_isa_counts.py x the issue rate measured in this run; the two
    # element-wise kernels are not counted: a lower bound) and HBM (algorithmic: 2k limbs read + 3k written per ciphertext; designed: what the four
    # kernels move by construction - extend R 2k W 2kb, squares R 2(k + kb) W 3(k + kb), floor R 3(k + kb) W 3k).  The per-kernel split of the same
    # launches is in profiles/r04_bench_kernel_trace_summary.txt.
    square = None
    if rank == 0:
        try:
            cts = 845
            t3 = g.ct_alloc(cts, 3)
            g.multiply(chans[0].h1, 0, chans[0].h1, 0, t3, 0, cts); g.sync()
            g.time_begin()
            for _ in range(5):
                g.multiply(chans[0].h1, 0, chans[0].h1, 0, t3, 0, cts)
            sq_ms = g.time_end() / 5
            g.free(t3)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import ks_isa_counts
            ss = ks_isa_counts.square_structure()
            kq, kb = g.k, g.get_option("aux_primes")
            waves = cts * (kq + kb) * 8                                      # 512-thread workgroups
            f_fp64 = waves * ss["fp64"]["per_thread"] / 1024 * fp64_ns * 1e-6
            f_valu = waves * (ss["fp64"]["per_thread"] * fp64_ns + ss["valu"]["per_thread"] * valu_ns) / 1024 * 1e-6
            limb = g.n * 8
            alg = cts * (2 * kq + 3 * kq) * limb
            designed = cts * (2 * kq + 2 * kb + 5 * (kq + kb) + 3 * (kq + kb) + 3 * kq) * limb
            square = {"chain": "Evaluator.Multiply(a, a) of 845 ciphertexts: k_behz_extend, k_square_fused (q: %d limbs, Bsk: %d limbs), k_behz_floor" % (kq, kb),
                      "ms_per_chain": round(sq_ms, 3), "share_of_batch": round(2 * sq_ms / (1e3 * dt / args.steps), 3),
                      "fp64_per_thread_square_fused": ss["fp64"]["per_thread"], "valu_other_per_thread_square_fused": ss["valu"]["per_thread"],
                      "counted_on": ss["kernel"] + " (the launch runs k_square_pipe for this batch size: the same two forward + three inverse transforms and tensor per "
                                    "block - 2 368 static FP64 instructions in both code objects - around a resident loop with the inverse roots in LDS)",
                      "fp64_ns_per_instr_in_situ": round(fp64_ns, 3), "fp64_issue_floor_in_situ_ms": round(f_fp64, 3), "frac_fp64_in_situ": round(f_fp64 / sq_ms, 3),
                      "valu_issue_floor_in_situ_ms": round(f_valu, 3), "frac_valu_in_situ": round(f_valu / sq_ms, 3),
                      "algorithmic_bytes": alg, "hbm_frac_algorithmic": round(alg / (sq_ms * 1e-3) / 8e12, 3),
                      "designed_bytes": designed, "hbm_frac_designed": round(designed / (sq_ms * 1e-3) / 8e12, 3),
                      "bound": "neither saturated: FP64 issue of the two transform kernels and the HBM traffic of the two element-wise kernels add up (the chain is four dependent launches)"}
        except Exception as ex:
            square = {"error": str(ex)[:300]}

    # ---- the reference's UNCHANGED caller: one evaluator call per ciphertext from the caller's threads (tools/replay_reference_calls.cpp), merged
    # by libcnhip's deferred submission; same inputs.  Main figure: the LITERAL pattern - a padded convolution tap is a fresh encryption of
    # the zero vector (PoolLayer.ElementAt, PoolLayer.cs:67-80: 645 per plaintext prime and batch, made on the device and queued like the
    # evaluator calls) - from Defaults.ThreadCount = all host cores (Defaults.cs), every decrypted slot checked against the integer model.
    # `skipped_taps`: padded taps passed as "no ciphertext" (what the batched path does): its final WORDS must equal the batched run's.
    unchanged = None
    if rank == 0 and world == 1 and not args.no_unchanged_caller:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import replay_reference_calls as rp
            ref_words = [ch.g.ct_download(ch.h5, 0, 10) for ch in chans]
            nthreads = args.caller_threads or effective_cores()[0]
            reps = max(2, min(args.steps, 20))           # (windows of up to 20 batches since round 6: filling and draining the queues of a window costs the unchanged caller ~2.5 ms - 0.5 ms per batch in a 5-batch window)
            # the SAME statistic on both sides of the ratio: best of five windows of `reps` steps for the batched path too (the headline
            # `value` stays the mean over all timed steps)
            bwin = []
            for _ in range(5):
                sync_all()
                tb = time.perf_counter()
                for _ in range(reps):
                    step()
                sync_all()
                bwin.append(1e3 * (time.perf_counter() - tb) / reps)
            batched_ms = min(bwin)
            # the other way to run the batched program: the two primes half a batch apart (cn_multiply + cn_relinearize, device-side ordering between the contexts) - the
            # default of rounds 3-6; the plain loop of `value` has cn_mul_relin pipeline each prime's squarings in parts instead
            pwin = []
            for _ in range(3):
                sync_all()
                tb = time.perf_counter()
                for _ in range(reps):
                    step_staggered(1) if not args.stagger else [ch.forward() for ch in chans]
                sync_all()
                pwin.append(1e3 * (time.perf_counter() - tb) / reps)
            plain_ms = min(pwin)
            # best of five (three for the secondary rows) short measurements each (a 100 ms window on a shared host: one scheduling hiccup is a third of it - the run-to-run
            # spread of single measurements is in profiles/r03_unchanged_caller_*.txt)
            lruns = [rp.measure(chans, layers, nthreads, reps, warmup=2, literal_taps=True) for _ in range(5)]
            lwin = [r[0] for r in lruns]
            lms, lwords = min(lruns, key=lambda r: r[0])
            dec = rp.decrypt_outputs(chans, lwords)
            lok = all(bool(np.array_equal(d, cm.model_mod_p_dense(x_int, layers, ch.g.t))) for d, ch in zip(dec, chans))
            ums, uwords = min((rp.measure(chans, layers, 4, reps, warmup=2) for _ in range(3)), key=lambda r: r[0])
            visible = None                                                 # os.cpu_count() threads where the cgroup grants fewer (a runtime that ignores the quota)
            if not args.caller_threads and effective_cores()[1] != nthreads:
                vms = min(rp.measure(chans, layers, effective_cores()[1], reps, warmup=2, literal_taps=True)[0] for _ in range(3))
                visible = {"threads": effective_cores()[1], "ms_per_step": round(vms, 2), "frac_of_batched": round(batched_ms / vms, 3)}
            locked = None                                                  # the same literal pattern with every deferred call under the context lock (defer = 1)
            try:
                kms = min(rp.measure(chans, layers, nthreads, reps, warmup=2, literal_taps=True, lockfree=False)[0] for _ in range(3))
                locked = {"threads": nthreads, "ms_per_step": round(kms, 2), "frac_of_batched": round(batched_ms / kms, 3),
                          "note": "the rounds 3-5 twin: every deferred call under the context lock, disposed arrays parked per thread and released 32 at a time (cn_free_many) - some zero "
                                  "vectors are still alive at the flush, nothing is folded; run behind the lock-free windows in the same process"}
            except Exception as ex:
                locked = {"error": str(ex)[:200]}
            unchanged = {"value": round(8192e3 / lms, 1), "unit": "images/s", "ms_per_step": round(lms, 2), "threads": nthreads,
                         "frac_of_batched": round(batched_ms / lms, 3), "verified_against_integer_model": lok, "verified_slots": 8192 * 10 * len(chans),
                         ("plain_loop_batched_ms" if args.stagger else "staggered_batched_ms"): round(plain_ms, 2), ("frac_of_plain_loop_batched" if args.stagger else "frac_of_staggered_batched"): round(plain_ms / lms, 3),
                         "timing": "best of 5 windows of %d steps on BOTH sides of frac_of_batched (skipped_taps, at_visible_cpu_count: best of 3)" % reps,
                         "windows_ms": {"unchanged": [round(x, 2) for x in lwin], "batched": [round(x, 2) for x in bwin]},
                         "frac_of_batched_mean_over_mean": round((sum(bwin) / len(bwin)) / (sum(lwin) / len(lwin)), 3), "at_visible_cpu_count": visible,
                         "pattern": "PoolLayer.Apply: per (map, corner) [cn_encrypt_zero_new per padded tap (the twin's one-call zero vector)] + cn_ct_alloc + cn_scalar_dot "
                                    "(K = 25 real handles) + cn_ct_alloc + cn_add_plain + cn_free of the product (at once: the queue folds the bias into the GEMM), ReleaseTemp: "
                                    "cn_free_many per 32 zero encryptions (CnDevice.DeferFree); SquareActivation: per column cn_mul_relin(count 1); BaseLayer.GetNext: cn_free_many "
                                    "of the layer's input columns; every ciphertext its own handle; cn_set_option(defer, 2): the deferrable calls are published to the context's "
                                    "submission ring without taking its lock and executed in claim order by whoever finds the lock free (round 6; `locked`: defer = 1, every call "
                                    "under the lock, rounds 2-5); at flush the scalar products of a layer are ONE launch per term count; threads = Defaults.ThreadCount = processor "
                                    "count (visible CPUs %d, cgroup quota %s)" % (effective_cores()[1], effective_cores()[2]),
                         "locked": locked,
                         "skipped_taps": {"value": round(8192e3 / ums, 1), "ms_per_step": round(ums, 2), "threads": 4, "frac_of_batched": round(batched_ms / ums, 3),
                                          "words_identical_to_batched": bool(all(np.array_equal(a, b) for a, b in zip(uwords, ref_words)))}}
        except Exception as ex:
            unchanged = {"error": str(ex)[:300]}

    # ---- NOT the headline, NOT the reference's call sequence: the same network with Relinearize moved behind the dense layers
    # (CryptoNetsChannel.forward_relinearize_late: the squarings leave size-3 products, Evaluator.MultiplyPlain / Add run on them, 110 key
    # switches per channel instead of 945).  Same SEAL operations on the same kernels, every slot checked against the integer model.
    late = None
    if rank == 0 and world == 1 and not args.no_relinearize_late:
        try:
            for _ in range(2):
                for ch in chans:
                    ch.forward_relinearize_late()
            sync_all()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                for ch in chans:
                    ch.forward_relinearize_late()
            sync_all()
            ldt = time.perf_counter() - t1
            ok = True
            for ch in chans:
                dh = ch.g.pt_alloc(10)
                ch.g.decrypt(ch.h5, 0, 10, dh, 0)
                ok = ok and bool(np.array_equal(np.ascontiguousarray(ch.g.decode_batch(dh, 0, 10).T), cm.model_mod_p_dense(x_int, layers, ch.g.t)))
                ch.g.free(dh)
            late = {"value": round(8192 * args.steps / ldt, 1), "unit": "images/s", "ms_per_step": round(1e3 * ldt / args.steps, 2),
                    "verified_against_integer_model": ok, "verified_slots": 8192 * 10 * len(chans), "key_switches_per_channel": 110,
                    "note": "opt-in program variant, not the reference's call sequence (PointwiseMultiply relinearizes at once: 945 key switches per "
                            "channel) and not the headline: ciphertext words differ from the reference's sequence, decrypted logits do not"}
        except Exception as ex:
            late = {"error": str(ex)[:300]}

    if rank == 0:
        images = 8192 * args.steps * world
        out = {"metric": "encrypted images/sec (CryptoNets-MNIST, N=8192)", "value": round(images / dt, 1), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
               "data": "synthetic images" + (", the reference's trained weights (CryptoNets/Weights.cs)" if args.weights == "trained" else ", synthetic weights"),
               "verified_against_integer_model": verified, "verified_slots": 8192 * 10 * len(chans), "logit_words_sha256": digest.hexdigest(),
               "launcher": os.environ.get("BENCH_LAUNCHER", "external" if "WORLD_SIZE" in os.environ else "none"), "process_group": "nccl" if dist is not None else None,
               "config": {"workload": "CryptoNets-MNIST 5-layer (conv 5x5 s2 x5 maps, square, dense 845->100, square, dense 100->10), "
                                      "8192-image batch per GPU per step, N=8192, 5 RNS limbs, plaintext primes {549764251649, 549764284417}, "
                                      "dbc=10; synthetic MNIST-like images encrypted on the device, inputs and keys resident in HBM; batched program with the zero-weighted padded "
                                      "convolution taps elided (the reference's literal call sequence: literal_call_sequence); weights: " + args.weights,
                          "program": ("the two plaintext primes half a batch apart (cn_multiply / cn_relinearize, device-side ordering between the contexts)" if args.stagger and not args.serialize else
                                      "five batched calls per plaintext prime, prime after prime; cn_mul_relin pipelines the 845 squarings in parts over two streams of its context"),
                          "padded_taps": "elided",     # the batched program of `value` skips the 645 x 2 zero-weighted padded convolution taps per batch the reference encrypts and
                                                       # multiplies inside its timed window (PoolLayer.cs:67-80); `literal_call_sequence` / `unchanged_caller` = the reference's literal sequence
                          "batch_per_gpu": 8192, "parallelism": "batch-sharded x%d, RCCL key broadcast only" % world,
                          "arithmetic": "exact modular integers over 43-49-bit RNS primes (results are u64 words, bit-identical to the integer "
                                        "oracle); products evaluated with error-free FP64 instruction sequences where the modulus is below 2^49, "
                                        "64-bit integer instructions otherwise"},
               # the figure for the reference's LITERAL call sequence (padded taps as fresh encryptions of zero, one call per ciphertext from Defaults.ThreadCount threads), next to `value`
               "literal_call_sequence": ({"value": unchanged.get("value"), "unit": "images/s", "ms_per_step": unchanged.get("ms_per_step"), "frac_of_value_program": unchanged.get("frac_of_batched"),
                                          "verified_against_integer_model": unchanged.get("verified_against_integer_model")} if isinstance(unchanged, dict) and "value" in unchanged else None),
               "roofline": roofline, "key_switch": key_switch, "square": square, "unchanged_caller": unchanged, "relinearize_late": late,
               # the WHOLE batch against HBM (inputs read once + outputs written once per layer, 640 KiB per ciphertext, per prime:
               # conv (784+845), square 845 x 2, dense (845+100), square 100 x 2, dense (100+10)): the path is FP64-issue bound, not HBM bound
               "batch_hbm": (lambda nbytes: {"algorithmic_bytes_per_step": nbytes, "achieved": round(nbytes / (dt / args.steps) / 1e9, 1), "peak": 8000.0,
                                             "unit": "GB/s", "frac": round(nbytes / (dt / args.steps) / 8e12, 4)})(
                   2 * (784 + 845 + 2 * 845 + 845 + 100 + 2 * 100 + 100 + 10) * 2 * g.k * g.n * 8)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(effective_cores()[0], layers)
        if world == 1 and not args.no_single_image:
            out["lola"], out["cifar"] = single_image_lines(local)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
