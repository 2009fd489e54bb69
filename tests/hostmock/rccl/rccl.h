// Stand-in for <rccl/rccl.h> (see ../hip/hip_runtime.h): a one-rank "communicator" whose collectives are identities.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclInvalidArgument = 4 };
typedef struct mockComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclMax = 2, ncclAvg = 4 } ncclRedOp_t;
extern "C" {
const char *ncclGetErrorString(ncclResult_t);
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclAllReduce(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclBroadcast(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
long mock_nccl_live_comms(void);
}
