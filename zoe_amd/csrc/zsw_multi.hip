// zsw_multi.hip — several GPUs behind one handle (SURVEY.md §8b "create(device_ids[], n)", §8e).
//
// The reference's many-readers affordance is SharedProfiles (src/alignment/profile_set.rs:552-560): profiles that any number
// of host threads may score against. The batched form here is the mirror image: one caller, a batch of reads, several GPUs.
// Alignments are independent, so the group shards the reads into contiguous index ranges [i*n/G, (i+1)*n/G), one per context,
// and drives every context from its own host thread; the reference and the scoring tables are replicated. There is no exchange
// during the computation. Results in host memory land in place (no collective at all); results in device memory are completed
// on every device by RCCL over xGMI: one in-place ncclAllGather per array when the shards are equal, grouped broadcasts when
// they differ by a read. One worker thread per context lives as long as the group.
//
// This file uses only the public C ABI (include/zoe_sw.h) and the HIP runtime; librccl is opened at the first device-memory
// call, so that hosts without it can still use everything else (the call then fails loudly with ZSW_ERR_UNSUPPORTED).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zoe_sw.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string* err) {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) {
            *err = std::string("librccl not found: ") + dlerror();
            return false;
        }
#define ZSW_SYM(field, name)                                         \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, name));     \
    if (!field) {                                                    \
        *err = std::string("librccl lacks ") + name;                 \
        return false;                                                \
    }
        ZSW_SYM(CommInitAll, "ncclCommInitAll")
        ZSW_SYM(CommDestroy, "ncclCommDestroy")
        ZSW_SYM(Broadcast, "ncclBroadcast")
        ZSW_SYM(AllGather, "ncclAllGather")
        ZSW_SYM(GroupStart, "ncclGroupStart")
        ZSW_SYM(GroupEnd, "ncclGroupEnd")
        ZSW_SYM(GetErrorString, "ncclGetErrorString")
#undef ZSW_SYM
        return true;
    }
};

}  // namespace

namespace {

// One worker per context, alive as long as the group: a call hands every worker its shard's task and waits for all of them.
struct Workers {
    std::mutex m;
    std::condition_variable wake, done;
    std::vector<std::thread> threads;
    std::vector<std::function<void()>> task;  // task[i] != nullptr: worker i has work
    int pending = 0;
    bool quit = false;

    void start(int n) {
        task.assign((size_t)n, nullptr);
        for (int i = 0; i < n; ++i)
            threads.emplace_back([this, i] {
                for (;;) {
                    std::function<void()> fn;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        wake.wait(lk, [&] { return quit || task[(size_t)i] != nullptr; });
                        if (quit) return;
                        fn = std::move(task[(size_t)i]);
                        task[(size_t)i] = nullptr;
                    }
                    fn();
                    {
                        std::lock_guard<std::mutex> lk(m);
                        --pending;
                    }
                    done.notify_all();
                }
            });
    }
    // runs fn(i) on worker i for every i, returns when all are finished (one caller at a time, as for a context)
    void run(const std::function<void(int)>& fn) {
        {
            std::lock_guard<std::mutex> lk(m);
            pending = (int)task.size();
            for (size_t i = 0; i < task.size(); ++i) task[i] = [&fn, i] { fn((int)i); };
        }
        wake.notify_all();
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return pending == 0; });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        wake.notify_all();
        for (auto& t : threads) t.join();
        threads.clear();
    }
};

}  // namespace

struct zsw_group {
    std::vector<int> devices;
    std::vector<zsw_context*> ctx;
    std::vector<ncclComm_t> comms;  // created with the first device-memory call
    Rccl rccl;
    Workers workers;                // started by zsw_group_create when the group has more than one context
    std::string err;
};

namespace {

// puts the calling thread's current device back when a call that visited the group's devices returns
struct RestoreDevice {
    int prev = -1;
    RestoreDevice() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~RestoreDevice() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

zsw_error gfail(zsw_group* g, zsw_error code, const std::string& what) {
    if (g) g->err = what;
    return code;
}

// shard i of n reads: [i*n/G, (i+1)*n/G)
void shard_of(uint64_t n, int i, int G, uint64_t* first, uint64_t* count) {
    const uint64_t a = (uint64_t)i * n / (uint64_t)G, b = (uint64_t)(i + 1) * n / (uint64_t)G;
    *first = a;
    *count = b - a;
}

// runs fn(i) for every context on its own host thread; returns the first error
template <typename Fn>
zsw_error for_each_context(zsw_group* g, Fn fn) {
    const int G = (int)g->ctx.size();
    std::vector<zsw_error> rc((size_t)G, ZSW_OK);
    if (G == 1) {
        rc[0] = fn(0);
    } else {
        g->workers.run([&](int i) { rc[(size_t)i] = fn(i); });
    }
    for (int i = 0; i < G; ++i)
        if (rc[(size_t)i] != ZSW_OK) {
            g->err = "context " + std::to_string(i) + " (device " + std::to_string(g->devices[(size_t)i]) + "): " +
                     zsw_last_error_string(g->ctx[(size_t)i]);
            return rc[(size_t)i];
        }
    return ZSW_OK;
}

}  // namespace

extern "C" {

zsw_error zsw_group_create(const int* device_ids, int n_devices, zsw_group** out) {
    if (!out) return ZSW_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return ZSW_ERR_INVALID_ARGUMENT;
    zsw_group* g = new (std::nothrow) zsw_group();
    if (!g) return ZSW_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n_devices; ++i) {
        zsw_context* c = nullptr;
        const zsw_error e = zsw_create(device_ids[i], &c);
        if (e != ZSW_OK) {  // zsw_last_error_string(NULL) holds the reason
            for (zsw_context* p : g->ctx) zsw_destroy(p);
            delete g;
            return e;
        }
        g->devices.push_back(device_ids[i]);
        g->ctx.push_back(c);
    }
    if (n_devices > 1) g->workers.start(n_devices);
    *out = g;
    return ZSW_OK;
}

void zsw_group_destroy(zsw_group* g) {
    if (!g) return;
    g->workers.stop();
    if (g->rccl.CommDestroy)
        for (ncclComm_t c : g->comms) (void)g->rccl.CommDestroy(c);
    for (zsw_context* p : g->ctx) zsw_destroy(p);
    delete g;
}

int zsw_group_size(const zsw_group* g) { return g ? (int)g->ctx.size() : 0; }

zsw_context* zsw_group_context(zsw_group* g, int i) { return (g && i >= 0 && i < (int)g->ctx.size()) ? g->ctx[(size_t)i] : nullptr; }

const char* zsw_group_last_error_string(const zsw_group* g) { return g ? g->err.c_str() : ""; }

zsw_error zsw_group_set_scoring(zsw_group* g, const int8_t* weights, int S, const uint8_t* index_map, int gap_open, int gap_extend) {
    if (!g) return ZSW_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const zsw_error e = zsw_set_scoring(g->ctx[i], weights, S, index_map, gap_open, gap_extend);
        if (e != ZSW_OK) return gfail(g, e, zsw_last_error_string(g->ctx[i]));
    }
    return ZSW_OK;
}

zsw_error zsw_group_set_reference(zsw_group* g, const uint8_t* reference, size_t len) {
    if (!g) return ZSW_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const zsw_error e = zsw_set_reference(g->ctx[i], reference, len, ZSW_MEM_HOST);
        if (e != ZSW_OK) return gfail(g, e, zsw_last_error_string(g->ctx[i]));
    }
    return ZSW_OK;
}

zsw_error zsw_group_score_batch_from(zsw_group* g, const zsw_batch* reads, int from_width, int preset_bits, uint32_t* out_score,
                                     uint8_t* out_status, uint8_t* out_tier) {
    if (!g) return ZSW_ERR_INVALID_ARGUMENT;
    if (!reads || !out_score || !out_status) return gfail(g, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    if (reads->mem != ZSW_MEM_HOST)
        return gfail(g, ZSW_ERR_INVALID_ARGUMENT, "zsw_group_score_batch_from takes host memory; device shards go to zsw_group_score_batch_from_device");
    const int G = (int)g->ctx.size();
    const uint64_t n = reads->n_reads;
    return for_each_context(g, [&](int i) -> zsw_error {
        uint64_t first, count;
        shard_of(n, i, G, &first, &count);
        if (count == 0) return ZSW_OK;
        zsw_batch b = *reads;
        b.n_reads = count;
        std::vector<uint64_t> rebased;
        if (reads->offsets) {  // the shard's bases start at offsets[first]
            rebased.resize(count + 1);
            const uint64_t base = reads->offsets[first];
            for (uint64_t k = 0; k <= count; ++k) rebased[k] = reads->offsets[first + k] - base;
            b.bases = reads->bases + base;
            b.offsets = rebased.data();
        } else {
            b.bases = reads->bases + first * (reads->encoding == ZSW_ENCODING_PACKED4 ? (uint64_t)(reads->fixed_len + 1) / 2 : (uint64_t)reads->fixed_len);
        }
        return zsw_score_batch_from(g->ctx[(size_t)i], &b, from_width, preset_bits, out_score + first, out_status + first,
                                    out_tier ? out_tier + first : nullptr, nullptr);
    });
}

}  // extern "C"

// Shared by the alignment entry points: one shard per context into shard-local arrays (sized from the count the library
// reports back), then the ciglets are laid out shard after shard in the caller's arrays and the records' offsets rebased.
namespace {

struct ShardAlign {
    std::vector<zsw_alignment> aln;
    std::vector<uint8_t> status, tier, op;
    std::vector<uint32_t> inc;
    uint64_t n_ciglets = 0;
};

template <typename Call>  // call(ctx, batch, aln, status, tier, inc, op, cap, n_ciglets)
zsw_error group_align(zsw_group* g, const zsw_batch* reads, zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier,
                      uint32_t* out_inc, uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets, Call call) {
    if (!reads || !out_aln || !out_status || !out_n_ciglets || (ciglet_cap && (!out_inc || !out_op)))
        return gfail(g, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    if (reads->mem != ZSW_MEM_HOST) return gfail(g, ZSW_ERR_INVALID_ARGUMENT, "the group alignment calls take host memory");
    const int G = (int)g->ctx.size();
    const uint64_t n = reads->n_reads;
    std::vector<ShardAlign> sh((size_t)G);
    zsw_error e = for_each_context(g, [&](int i) -> zsw_error {
        uint64_t first, count;
        shard_of(n, i, G, &first, &count);
        if (count == 0) return ZSW_OK;
        zsw_batch b = *reads;
        b.n_reads = count;
        std::vector<uint64_t> rebased;
        if (reads->offsets) {
            rebased.resize(count + 1);
            const uint64_t base = reads->offsets[first];
            for (uint64_t k = 0; k <= count; ++k) rebased[k] = reads->offsets[first + k] - base;
            b.bases = reads->bases + base;
            b.offsets = rebased.data();
        } else {
            b.bases = reads->bases + first * (reads->encoding == ZSW_ENCODING_PACKED4 ? (uint64_t)(reads->fixed_len + 1) / 2 : (uint64_t)reads->fixed_len);
        }
        ShardAlign& s = sh[(size_t)i];
        s.aln.resize(count);
        s.status.resize(count);
        s.tier.resize(count);
        uint64_t cap = 4 * count + 16;
        for (;;) {
            s.inc.resize(cap);
            s.op.resize(cap);
            const zsw_error rc = call(g->ctx[(size_t)i], &b, s.aln.data(), s.status.data(), s.tier.data(), s.inc.data(), s.op.data(), cap, &s.n_ciglets);
            if (rc == ZSW_ERR_INVALID_ARGUMENT && s.n_ciglets > cap) {  // capacity too small: the required size came back
                cap = s.n_ciglets;
                continue;
            }
            return rc;
        }
    });
    if (e != ZSW_OK) return e;
    uint64_t total = 0;
    for (const ShardAlign& s : sh) total += s.n_ciglets;
    *out_n_ciglets = total;
    if (total > ciglet_cap) return gfail(g, ZSW_ERR_INVALID_ARGUMENT, "ciglet capacity too small; required size returned");
    uint64_t base = 0;
    for (int i = 0; i < G; ++i) {
        uint64_t first, count;
        shard_of(n, i, G, &first, &count);
        const ShardAlign& s = sh[(size_t)i];
        for (uint64_t k = 0; k < count; ++k) {
            zsw_alignment a = s.aln[k];
            a.ciglet_offset += base;
            out_aln[first + k] = a;
            out_status[first + k] = s.status[k];
            if (out_tier) out_tier[first + k] = s.tier[k];
        }
        for (uint64_t k = 0; k < s.n_ciglets; ++k) {
            out_inc[base + k] = s.inc[k];
            out_op[base + k] = s.op[k];
        }
        base += s.n_ciglets;
    }
    return ZSW_OK;
}

}  // namespace

extern "C" {

zsw_error zsw_group_align_batch_from(zsw_group* g, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                                     zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                                     uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets) {
    if (!g) return ZSW_ERR_INVALID_ARGUMENT;
    return group_align(g, reads, out_aln, out_status, out_tier, out_inc, out_op, ciglet_cap, out_n_ciglets,
                       [&](zsw_context* c, const zsw_batch* b, zsw_alignment* aln, uint8_t* st, uint8_t* tier, uint32_t* inc, uint8_t* op,
                           uint64_t cap, uint64_t* nc) {
                           return zsw_align_batch_from(c, b, from_width, preset_bits, invert, aln, st, tier, inc, op, cap, nc, nullptr);
                       });
}

zsw_error zsw_group_align_3pass_batch_from(zsw_group* g, const zsw_batch* reads, int from_width, int preset_bits, int invert,
                                           zsw_alignment* out_aln, uint8_t* out_status, uint8_t* out_tier, uint32_t* out_inc,
                                           uint8_t* out_op, uint64_t ciglet_cap, uint64_t* out_n_ciglets) {
    if (!g) return ZSW_ERR_INVALID_ARGUMENT;
    return group_align(g, reads, out_aln, out_status, out_tier, out_inc, out_op, ciglet_cap, out_n_ciglets,
                       [&](zsw_context* c, const zsw_batch* b, zsw_alignment* aln, uint8_t* st, uint8_t* tier, uint32_t* inc, uint8_t* op,
                           uint64_t cap, uint64_t* nc) {
                           return zsw_align_3pass_batch_from(c, b, from_width, preset_bits, invert, aln, st, tier, inc, op, cap, nc, nullptr);
                       });
}

zsw_error zsw_group_score_batch_from_device(zsw_group* g, const zsw_batch* shards, int from_width, int preset_bits,
                                            uint32_t* const* out_score, uint8_t* const* out_status) {
    if (!g) return ZSW_ERR_INVALID_ARGUMENT;
    if (!shards || !out_score || !out_status) return gfail(g, ZSW_ERR_INVALID_ARGUMENT, "null argument");
    const int G = (int)g->ctx.size();
    RestoreDevice restore;
    std::vector<uint64_t> first((size_t)G + 1, 0);
    for (int i = 0; i < G; ++i) {
        if (shards[i].mem != ZSW_MEM_DEVICE || !out_score[i] || !out_status[i])
            return gfail(g, ZSW_ERR_INVALID_ARGUMENT, "every shard and its output arrays must be device memory of its context's GPU");
        first[(size_t)i + 1] = first[(size_t)i] + shards[i].n_reads;
    }
    if (g->comms.empty()) {  // one communicator per context, created once
        // RCCL wants one rank per GPU: the host-memory entry points accept several contexts on one device, this one does not
        for (int i = 0; i < G; ++i)
            for (int j = i + 1; j < G; ++j)
                if (g->devices[(size_t)i] == g->devices[(size_t)j])
                    return gfail(g, ZSW_ERR_INVALID_ARGUMENT, "zsw_group_score_batch_from_device needs distinct devices (device " +
                                                                  std::to_string(g->devices[(size_t)i]) + " appears twice)");
        if (!g->rccl.load(&g->err)) return ZSW_ERR_UNSUPPORTED;
        g->comms.resize((size_t)G);
        const ncclResult_t r = g->rccl.CommInitAll(g->comms.data(), G, g->devices.data());
        if (r != ncclSuccess) {
            g->comms.clear();
            return gfail(g, ZSW_ERR_HIP, std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r));
        }
    }
    // every context scores its shard straight into its slice of its device's result arrays ...
    zsw_error e = for_each_context(g, [&](int i) -> zsw_error {
        if (shards[i].n_reads == 0) return ZSW_OK;
        const zsw_error rc = zsw_score_batch_from(g->ctx[(size_t)i], &shards[i], from_width, preset_bits, out_score[i] + first[(size_t)i],
                                                  out_status[i] + first[(size_t)i], nullptr, nullptr);
        if (rc != ZSW_OK) return rc;
        return (hipSetDevice(g->devices[(size_t)i]) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess) ? ZSW_OK : ZSW_ERR_HIP;
    });
    if (e != ZSW_OK) return e;
    // ... and one grouped collective completes the arrays on every device, in place. Equal shards: one ncclAllGather per array
    // (rank r's piece already sits at r * count of its own arrays). Shards that differ by a read: shard r is broadcast from rank r.
    // An error inside the bracket only ends the loop: ncclGroupEnd always runs, or the communicators would stay in an open group.
    bool equal = true;
    for (int i = 1; i < G; ++i) equal = equal && shards[i].n_reads == shards[0].n_reads;
    bool device_error = false;
    ncclResult_t r = g->rccl.GroupStart();
    if (equal && shards[0].n_reads > 0) {
        const uint64_t cnt = shards[0].n_reads;
        for (int i = 0; i < G && r == ncclSuccess && !device_error; ++i) {
            if (hipSetDevice(g->devices[(size_t)i]) != hipSuccess) {
                device_error = true;
                break;
            }
            r = g->rccl.AllGather(out_score[i] + (uint64_t)i * cnt, out_score[i], cnt, ncclUint32, g->comms[(size_t)i], nullptr);
            if (r == ncclSuccess) r = g->rccl.AllGather(out_status[i] + (uint64_t)i * cnt, out_status[i], cnt, ncclUint8, g->comms[(size_t)i], nullptr);
        }
    } else {
        for (int root = 0; root < G && r == ncclSuccess && !device_error; ++root) {
            const uint64_t cnt = shards[root].n_reads;
            if (cnt == 0) continue;
            for (int i = 0; i < G && r == ncclSuccess; ++i) {
                if (hipSetDevice(g->devices[(size_t)i]) != hipSuccess) {
                    device_error = true;
                    break;
                }
                uint32_t* ps = out_score[i] + first[(size_t)root];
                uint8_t* pt = out_status[i] + first[(size_t)root];
                r = g->rccl.Broadcast(ps, ps, cnt, ncclUint32, root, g->comms[(size_t)i], nullptr);
                if (r == ncclSuccess) r = g->rccl.Broadcast(pt, pt, cnt, ncclUint8, root, g->comms[(size_t)i], nullptr);
            }
        }
    }
    const ncclResult_t r2 = g->rccl.GroupEnd();
    if (device_error) return gfail(g, ZSW_ERR_HIP, "hipSetDevice inside the RCCL gather");
    if (r != ncclSuccess || r2 != ncclSuccess)
        return gfail(g, ZSW_ERR_HIP, std::string("RCCL gather: ") + g->rccl.GetErrorString(r != ncclSuccess ? r : r2));
    for (int i = 0; i < G; ++i)
        if (hipSetDevice(g->devices[(size_t)i]) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)
            return gfail(g, ZSW_ERR_HIP, "synchronising the gather");
    return ZSW_OK;
}

}  // extern "C"
