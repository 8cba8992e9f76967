// comm.hip — the path's one exchange step, by the device library itself: the film rows a rank owns travel to one rank over RCCL
// (xGMI inside a node). BASELINE.json north_star: "the framebuffer is tile-sharded across GPUs with an RCCL gather over xGMI only for
// final accumulation"; the reference has no collective to match (its devices render whole films: IRenderDevice::render,
// src/runtime/device/IRenderDevice.h:44).
//
// librccl.so is opened at run time (dlopen): a single-GPU process never loads it, and the library has no link-time dependency on it.
// No torch in this path: a launcher only has to hand every rank the 128-byte id rank 0 made (ignis_amd/comm.py: a TCP exchange on
// MASTER_ADDR / MASTER_PORT + 1, the variables `python -m torch.distributed.run` and ignis_amd.cli's own launcher export).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "comm.h"

namespace igdev {

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*)                                                                   = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int)                                             = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*)                                                            = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t)                                                                      = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                       = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                             = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)   = nullptr;
    ncclResult_t (*GroupStart)()                                                                                 = nullptr;
    ncclResult_t (*GroupEnd)()                                                                                   = nullptr;
    const char* (*GetErrorString)(ncclResult_t)                                                                  = nullptr;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r = [] {
        Rccl x;
        // (a process that imported torch first has torch's bundled librccl loaded already: the plain name resolves to it, and to
        // /opt/rocm/lib otherwise through this library's rpath)
        for (const char* name : { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1" }) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.lib)
                break;
        }
        if (!x.lib) {
            x.error = std::string("librccl.so could not be loaded: ") + dlerror();
            return x;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(x.lib, n);
            if (!p && x.error.empty())
                x.error = std::string("librccl.so has no symbol ") + n;
            return p;
        };
        x.GetUniqueId    = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank   = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommCount      = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
        x.CommDestroy    = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.Send           = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
        x.Recv           = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
        x.AllReduce      = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
        x.GroupStart     = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd       = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        return x;
    }();
    return r;
}

void need(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess)
        throw CommError{ std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error") };
}
void need(hipError_t e, const char* what)
{
    if (e != hipSuccess)
        throw CommError{ std::string(what) + ": " + hipGetErrorString(e) };
}
Rccl& loaded()
{
    Rccl& r = rccl();
    if (!r.error.empty())
        throw CommError{ r.error };
    return r;
}

} // namespace

static_assert(sizeof(ncclUniqueId) == kCommIdBytes, "igd_comm_unique_id's id is an ncclUniqueId");

bool comm_available(std::string& why)
{
    const Rccl& r = rccl();
    why           = r.error;
    return r.error.empty();
}

void comm_unique_id(uint8_t* id)
{
    ncclUniqueId u;
    need(loaded().GetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, &u, sizeof(u));
}

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    float* stage       = nullptr; // packed rows: one rank's worth on a sender, world ranks' worth on the receiver
    size_t stage_bytes = 0;
    double* scratch    = nullptr; // igd_comm_allreduce_f64
    size_t scratch_count = 0;
};

Comm* comm_create(const uint8_t* id, int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world)
        throw CommError{ "igd_comm_init: rank / world out of range" };
    Rccl& r = loaded();
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    Comm* c  = new Comm;
    c->rank  = rank;
    c->world = world;
    const ncclResult_t res = r.CommInitRank(&c->comm, world, u, rank); // (on the calling thread's current device: igd_comm_init selects it)
    if (res != ncclSuccess) {
        delete c;
        need(res, "ncclCommInitRank");
    }
    return c;
}

void comm_destroy(Comm* c)
{
    if (!c)
        return;
    if (c->comm)
        (void)rccl().CommDestroy(c->comm);
    if (c->stage)
        (void)hipFree(c->stage);
    if (c->scratch)
        (void)hipFree(c->scratch);
    delete c;
}

int comm_world_size(const Comm* c)
{
    int n = 0;
    need(loaded().CommCount(c->comm, &n), "ncclCommCount");
    return n;
}

static size_t rowsOf(int rank, int world, int height) { return rank < height ? (size_t)((height - rank + world - 1) / world) : 0; }

// Rows rank, rank + world, ... of fb [height][width][3] hold this rank's shard. Every rank but dst packs its rows (a strided
// device-to-device copy) and sends them; dst receives every shard next to each other and scatters the rows into its framebuffer.
// Bytes on the wire per rank: ceil(height / world) x width x 12 — what a reduce(SUM) of zero-padded films moves `world` times.
// loopback (tests on one GPU): dst sends its own rows to itself through the same calls, clears them and puts them back.
void comm_gather_rows(Comm* c, float* fb, int width, int height, int dst, hipStream_t stream, bool loopback)
{
    Rccl& r = loaded();
    if (dst < 0 || dst >= c->world)
        throw CommError{ "igd_comm_gather_rows: destination rank out of range" };
    if (c->world == 1 && !loopback)
        return;
    const size_t row_bytes = (size_t)width * 3 * sizeof(float);
    const size_t rows_max  = rowsOf(0, c->world, height);
    const size_t slots     = c->rank == dst ? (size_t)c->world : 1;
    const size_t want      = std::max<size_t>(1, rows_max * row_bytes * slots);
    if (c->stage_bytes < want) {
        if (c->stage)
            need(hipFree(c->stage), "hipFree");
        c->stage = nullptr, c->stage_bytes = 0;
        need(hipMalloc(reinterpret_cast<void**>(&c->stage), want), "hipMalloc (row staging)");
        c->stage_bytes = want;
    }
    auto slotOf = [&](int rank) { return reinterpret_cast<uint8_t*>(c->stage) + (c->rank == dst ? (size_t)rank : 0) * rows_max * row_bytes; };
    auto pack   = [&](int rank, uint8_t* to) {
        const size_t n = rowsOf(rank, c->world, height);
        if (n)
            need(hipMemcpy2DAsync(to, row_bytes, reinterpret_cast<uint8_t*>(fb) + (size_t)rank * row_bytes, row_bytes * (size_t)c->world, row_bytes, n, hipMemcpyDeviceToDevice, stream), "hipMemcpy2DAsync (pack rows)");
    };
    auto unpack = [&](int rank, const uint8_t* from) {
        const size_t n = rowsOf(rank, c->world, height);
        if (n)
            need(hipMemcpy2DAsync(reinterpret_cast<uint8_t*>(fb) + (size_t)rank * row_bytes, row_bytes * (size_t)c->world, from, row_bytes, row_bytes, n, hipMemcpyDeviceToDevice, stream), "hipMemcpy2DAsync (unpack rows)");
    };
    if (c->rank != dst) {
        const size_t n = rowsOf(c->rank, c->world, height);
        pack(c->rank, slotOf(c->rank));
        if (n)
            need(r.Send(c->stage, n * (size_t)width * 3, ncclFloat32, dst, c->comm, stream), "ncclSend");
    } else {
        uint8_t* own_copy = nullptr;
        if (loopback) {
            // this rank's rows out through a send to itself, and their place in the film cleared, so that what comes back is what travelled
            const size_t n = rowsOf(dst, c->world, height);
            need(hipMalloc(reinterpret_cast<void**>(&own_copy), std::max<size_t>(1, n * row_bytes)), "hipMalloc (loopback)");
            pack(dst, own_copy);
            if (n)
                need(hipMemset2DAsync(reinterpret_cast<uint8_t*>(fb) + (size_t)dst * row_bytes, row_bytes * (size_t)c->world, 0, row_bytes, n, stream), "hipMemset2DAsync");
        }
        need(r.GroupStart(), "ncclGroupStart");
        for (int k = 0; k < c->world; ++k) {
            const size_t n = rowsOf(k, c->world, height);
            if (!n || (k == dst && !loopback))
                continue;
            if (k == dst)
                need(r.Send(own_copy, n * (size_t)width * 3, ncclFloat32, dst, c->comm, stream), "ncclSend (loopback)");
            need(r.Recv(slotOf(k), n * (size_t)width * 3, ncclFloat32, k, c->comm, stream), "ncclRecv");
        }
        need(r.GroupEnd(), "ncclGroupEnd");
        for (int k = 0; k < c->world; ++k)
            if (k != dst || loopback)
                unpack(k, slotOf(k));
        if (own_copy) {
            need(hipStreamSynchronize(stream), "hipStreamSynchronize");
            need(hipFree(own_copy), "hipFree");
        }
    }
    need(hipStreamSynchronize(stream), "hipStreamSynchronize (gather)");
}

// values[i] <- op over the ranks of values[i] (op 0 sum, 2 max): the statistics / the slowest rank's clock of a sharded run, and,
// being a collective every rank waits in, its barrier
void comm_allreduce_f64(Comm* c, double* values, int count, int op, hipStream_t stream)
{
    Rccl& r = loaded();
    if (count <= 0)
        return;
    if (op != 0 && op != 2)
        throw CommError{ "igd_comm_allreduce_f64: op must be 0 (sum) or 2 (max)" };
    if (c->scratch_count < (size_t)count) {
        if (c->scratch)
            need(hipFree(c->scratch), "hipFree");
        c->scratch = nullptr, c->scratch_count = 0;
        need(hipMalloc(reinterpret_cast<void**>(&c->scratch), (size_t)count * sizeof(double)), "hipMalloc (all-reduce)");
        c->scratch_count = (size_t)count;
    }
    need(hipMemcpyAsync(c->scratch, values, (size_t)count * sizeof(double), hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
    need(r.AllReduce(c->scratch, c->scratch, (size_t)count, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->comm, stream), "ncclAllReduce");
    need(hipMemcpyAsync(values, c->scratch, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    need(hipStreamSynchronize(stream), "hipStreamSynchronize (all-reduce)");
}

} // namespace igdev
