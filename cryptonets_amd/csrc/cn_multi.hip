// libcnhip.so host runtime (5/5): evaluation keys of one context to every other (RCCL / peer copies) - the only exchange of the multi-GPU path.
#include "cn_api_shared.h"

// ---------------------------------------------------------------- multi-GPU: evaluation keys of ctxs[0] to every other context
// The path shards by independent batches / plaintext primes: the only exchange is this one-time key broadcast.  A host that
// runs one process per GPU (bench.py) broadcasts with torch.distributed and adopts the buffers (cn_set_relin_key, is_device_ptr = 1); a
// single-process multi-threaded host (the C# one) calls this: contexts on OTHER devices receive the keys with ONE RCCL broadcast per key
// over xGMI (librccl is loaded on demand; without it: peer copies), contexts on the root's device with device-to-device copies.
#include <dlfcn.h>
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast"); GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};
const int kNcclUint64 = 5;          // ncclDataType_t
}
#define NCCLCHK(x) do { int r_ = (x); if (r_) { rc = fail(CN_ERR_HIP, "%s failed: %s", #x, R.GetErrorString ? R.GetErrorString(r_) : "rccl error"); goto done; } } while (0)
extern "C" int cn_ctx_broadcast_keys(cn_ctx **ctxs, int n) {
    if (!ctxs || n < 1 || !ctxs[0]) return fail(CN_ERR_ARG, "null argument");
    cn_ctx *root = ctxs[0];
    for (int i = 1; i < n; i++) {
        cn_ctx *c = ctxs[i];
        if (!c || c == root) return fail(CN_ERR_ARG, "context %d is null or the root itself", i);
        bool same = c->hc.n == root->hc.n && c->hc.k == root->hc.k && c->hc.t.q == root->hc.t.q && c->hc.dbc == root->hc.dbc && c->hc.gdbc == root->hc.gdbc;
        for (uint32_t j = 0; same && j < root->hc.k; j++) same = c->hc.q[j].q == root->hc.q[j].q;
        if (!same) return fail(CN_ERR_ARG, "context %d has other encryption parameters than the root", i);
    }
    // everything in flight on the contexts is finished first; then the ROOT's lock is held for the whole broadcast (its key table is
    // read and its key buffers are the sources: a concurrent cn_set_galois_key on the root must not free or re-map them meanwhile).
    // Callers broadcast at start-up, from one thread, with one root.
    for (int i = 0; i < n; i++) CHECK(cn_sync(ctxs[i]));
    CnGuard root_lock(root->mu);
    struct Item { uint64_t elt; bool galois; KsKey src; size_t words; };
    std::vector<Item> items;
    if (root->rlk.d) items.push_back({0, false, root->rlk, cn_key_words(root, 0)});
    for (auto &kv : root->gk) if (kv.second.d) items.push_back({kv.first, true, kv.second, cn_key_words(root, 1)});
    if (items.empty()) return fail(CN_ERR_NOKEY, "the root context has no evaluation keys");
    // destination buffers
    std::vector<std::vector<uint64_t *>> dst(n, std::vector<uint64_t *>(items.size(), nullptr));
    int rc = 0;
    Rccl R;
    std::vector<int> devs;                   // distinct devices, the root's first; leader[d] = first context on devs[d]
    std::vector<int> leader;
    std::vector<void *> comms;
    auto dev_index = [&](int device) { for (size_t d = 0; d < devs.size(); d++) if (devs[d] == device) return (int)d; return -1; };
    for (int i = 0; i < n; i++) if (dev_index(ctxs[i]->device) < 0) { devs.push_back(ctxs[i]->device); leader.push_back(i); }
    for (int i = 1; i < n && !rc; i++) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) { rc = fail(CN_ERR_HIP, "hipSetDevice failed"); break; }
        for (size_t x = 0; x < items.size(); x++)
            if (hipMalloc((void **)&dst[i][x], items[x].words * 8) != hipSuccess) { rc = fail(CN_ERR_HIP, "out of device memory for the broadcast keys"); break; }
    }
    const bool force = getenv("CN_BCAST_FORCE_RCCL") && atoi(getenv("CN_BCAST_FORCE_RCCL"));
    const bool use_rccl = !rc && (devs.size() > 1 || force) && R.load();
    if (!rc && use_rccl) {
        comms.assign(devs.size(), nullptr);
        NCCLCHK(R.CommInitAll(comms.data(), (int)devs.size(), devs.data()));
        for (size_t x = 0; x < items.size(); x++) {
            NCCLCHK(R.GroupStart());
            for (size_t d = 0; d < devs.size(); d++) {
                cn_ctx *c = ctxs[leader[d]];
                void *buf = d == 0 ? (void *)items[x].src.d : (void *)dst[leader[d]][x];
                if (hipSetDevice(c->device) != hipSuccess) { rc = fail(CN_ERR_HIP, "hipSetDevice failed"); goto done; }
                NCCLCHK(R.Broadcast(buf, buf, items[x].words, kNcclUint64, 0, comms[d], c->stream));
            }
            NCCLCHK(R.GroupEnd());
        }
    }
    for (int i = 1; i < n && !rc; i++) {     // contexts that did not receive through RCCL: copies from their device's leader (or from the root)
        const int d = dev_index(ctxs[i]->device);
        const bool got = use_rccl && leader[d] == i;
        if (got) continue;
        const int from = (use_rccl || d == 0) ? leader[d] : 0;
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) { rc = fail(CN_ERR_HIP, "hipSetDevice failed"); break; }
        if (from != 0 && hipStreamSynchronize(ctxs[from]->stream) != hipSuccess) { rc = fail(CN_ERR_HIP, "synchronisation failed"); break; }
        for (size_t x = 0; x < items.size() && !rc; x++) {
            const void *src = from == 0 ? (const void *)items[x].src.d : (const void *)dst[from][x];
            hipError_t e = ctxs[from]->device == ctxs[i]->device ? hipMemcpyAsync(dst[i][x], src, items[x].words * 8, hipMemcpyDeviceToDevice, ctxs[i]->stream)
                                                                 : hipMemcpyPeerAsync(dst[i][x], ctxs[i]->device, src, ctxs[from]->device, items[x].words * 8, ctxs[i]->stream);
            if (e != hipSuccess) rc = fail(CN_ERR_HIP, "key copy failed: %s", hipGetErrorString(e));
        }
    }
done:
    for (int i = 0; i < n; i++) { (void)hipSetDevice(ctxs[i]->device); (void)hipStreamSynchronize(ctxs[i]->stream); }
    for (void *cm : comms) if (cm) (void)R.CommDestroy(cm);
    for (int i = 1; i < n; i++) {
        CnGuard lk(ctxs[i]->mu);
        // the keys are of ONE decomposition convention (cn_set_option("ks_xi"), settled per context by the client's start-up self-test): a replica that adopts the
        // root's keys adopts its convention with them - with the other one every Relinearize / Rotate would return rc 0 and garbage
        if (!rc && ctxs[i]->hc.ks_xi != root->hc.ks_xi) {
            if (ctxs[i]->capturing || ctxs[i]->graphs_alive) rc = fail(CN_ERR_ARG, "context %d holds recorded graphs of the other key-switch convention", i);
            else {
                (void)hipSetDevice(ctxs[i]->device);
                ctxs[i]->hc.ks_xi = root->hc.ks_xi;
                if (hipMemcpy(ctxs[i]->dc, &ctxs[i]->hc, sizeof(DevConsts), hipMemcpyHostToDevice) != hipSuccess) rc = fail(CN_ERR_HIP, "constant upload failed on context %d", i);
                // the keys this replica already holds were made for the OTHER convention: those the broadcast does not overwrite are dropped (a rotation by one of
                // their elements then fails with CN_ERR_NOKEY instead of returning rc 0 and garbage - ADVICE r05)
                if (!rc) {
                    auto carried = [&](bool galois, uint64_t elt) { for (const Item &it : items) if (it.galois == galois && (!galois || it.elt == elt)) return true; return false; };
                    if (ctxs[i]->rlk.d && !carried(false, 0)) { if (ctxs[i]->rlk.owned) (void)hipFree(ctxs[i]->rlk.d); ctxs[i]->rlk = KsKey{nullptr, false, false}; }
                    for (auto it = ctxs[i]->gk.begin(); it != ctxs[i]->gk.end();) {
                        if (!carried(true, it->first)) { if (it->second.owned && it->second.d) (void)hipFree(it->second.d); it = ctxs[i]->gk.erase(it); } else ++it;
                    }
                }
            }
        }
        for (size_t x = 0; x < items.size(); x++) {
            if (!dst[i][x]) continue;
            if (rc) { (void)hipSetDevice(ctxs[i]->device); (void)hipFree(dst[i][x]); continue; }
            KsKey &slot = items[x].galois ? ctxs[i]->gk[items[x].elt] : ctxs[i]->rlk;
            if (slot.owned && slot.d) { (void)hipSetDevice(ctxs[i]->device); (void)hipFree(slot.d); }
            slot = KsKey{dst[i][x], true, items[x].src.f64};           // the words arrive in the form the root keeps them (FP64 image or u64) ...
            const bool want = keys_as_f64(ctxs[i]);                    // ... and are converted when this context keeps the other form (f64 = 0, legacy_ntt)
            if (want != slot.f64) {
                (void)hipSetDevice(ctxs[i]->device);
                const size_t words = items[x].words;
                if (want) hipLaunchKernelGGL(k_u64_to_f64, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctxs[i]->stream, slot.d, words);
                else hipLaunchKernelGGL(k_f64_to_u64, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctxs[i]->stream, slot.d, words);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctxs[i]->stream) != hipSuccess) rc = fail(CN_ERR_HIP, "key conversion failed on context %d", i);
                slot.f64 = want;
            }
        }
    }
    (void)hipSetDevice(root->device);
    return rc;
}
