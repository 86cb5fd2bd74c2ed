// kifs_comm.hpp -- RCCL as the library uses it: the handful of entry points of the one-process, N-device
// gather (kifs_multi.cpp), resolved from librccl.so.1 on first use.  A host that drives one GPU never loads
// RCCL; a host process that already holds a librccl.so.1 (PyTorch-ROCm ships its own) shares that copy.
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace kifs {
namespace host {

struct RcclApi {
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    int version = 0;
};

// The process-wide table, or nullptr when librccl cannot be opened or lacks a symbol (KIFS_DEBUG=1 says which).
const RcclApi* rccl();
// KIFS_DEBUG=1 prints the failing RCCL call to stderr.
bool nccl_ok(ncclResult_t r, const char* what);

}  // namespace host
}  // namespace kifs
