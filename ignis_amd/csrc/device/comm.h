// comm.h — the RCCL communicator of a tile-sharded render (comm.hip), as device.hip's C entry points igd_comm_* see it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>

namespace igdev {

constexpr int kCommIdBytes = 128; // sizeof(ncclUniqueId)

struct CommError : std::runtime_error {
    explicit CommError(const std::string& m)
        : std::runtime_error(m)
    {
    }
};

struct Comm;
bool comm_available(std::string& why);                    // librccl.so loaded with every symbol comm.hip calls
void comm_unique_id(uint8_t* id);                         // ncclGetUniqueId
Comm* comm_create(const uint8_t* id, int rank, int world); // ncclCommInitRank on the current device
void comm_destroy(Comm* c);
int comm_world_size(const Comm* c);                        // ncclCommCount
void comm_gather_rows(Comm* c, float* fb, int width, int height, int dst, hipStream_t stream, bool loopback);
void comm_allreduce_f64(Comm* c, double* values, int count, int op, hipStream_t stream);

} // namespace igdev
