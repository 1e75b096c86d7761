// pg_comm_rccl.cpp -- device-to-device block exchange with RCCL send/recv over xGMI.
//
// The reference moves every block D2H -> MPI_Isend -> MPI_Recv -> H2D (…0201000.cu:196-201,
// src/pangulu_communication.c:1809,1850,1943).  Here a finished block record goes straight from the owner's HBM
// into a receive slot in the consumer's HBM.  RCCL has no probe and no any-source receive, so:
//   * the TCP control plane of SocketComm announces every block (frame + 32-byte record header) on the ordered
//     per-pair socket; the receiver learns identity and size from it, takes a slot and posts the matching ncclRecv;
//   * xGMI is point-to-point (one link per GPU pair), so there is ONE 2-rank communicator PER ORDERED PAIR
//     (i -> j), used by i only for ncclSend and by j only for ncclRecv, each on its own HIP stream.  Every
//     communicator therefore sees a strictly FIFO sequence that is identical on both sides (announce order), no
//     communicator is ever used from two streams, and traffic between different pairs cannot block each other --
//     which a single 8-rank communicator with one send and one receive stream per rank would (a cycle of
//     head-of-line-blocked sends).
// librccl.so is loaded lazily so the library (and every CPU test) works where RCCL is absent.
// Before first use every pair exchanges a test pattern in both directions; if initialisation or that self test
// does not finish within PANGULU_AMD_RCCL_TIMEOUT_S (default 180 s) on any rank, ALL ranks fall back to the
// host-staged path together and say so.
#include <dlfcn.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <future>

#include "pg_comm_socket.h"

namespace pg
{

namespace
{

struct NcclUniqueId
{
    char internal[128];
};
typedef void *ncclComm_t;

struct RcclApi
{
    void *h = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, NcclUniqueId, int) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load()
    {
        if (h)
            return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names)
        {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h)
                break;
        }
        if (!h)
            return false;
        GetUniqueId = (int (*)(NcclUniqueId *))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (int (*)(ncclComm_t *, int, NcclUniqueId, int))dlsym(h, "ncclCommInitRank");
        Send = (int (*)(const void *, size_t, int, int, ncclComm_t, hipStream_t))dlsym(h, "ncclSend");
        Recv = (int (*)(void *, size_t, int, int, ncclComm_t, hipStream_t))dlsym(h, "ncclRecv");
        CommDestroy = (int (*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
        GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
        return GetUniqueId && CommInitRank && Send && Recv && CommDestroy;
    }
};
RcclApi R;
const int NCCL_CHAR = 0; // ncclInt8 / ncclChar

#define HIPC(expr)                                                                               \
    do                                                                                           \
    {                                                                                            \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            fatal("HIP error %s at %s:%d", hipGetErrorString(e_), __FILE__, __LINE__);           \
    } while (0)
#define NCCLC(expr)                                                                              \
    do                                                                                           \
    {                                                                                            \
        int r_ = (expr);                                                                         \
        if (r_ != 0)                                                                             \
            fatal("RCCL error %s at %s:%d", R.GetErrorString ? R.GetErrorString(r_) : "?", __FILE__, __LINE__); \
    } while (0)

enum : int
{
    TAG_NCCL_ID = 0x7fffff00,
    TAG_RCCL_OK = 0x7fffff01
};

struct RcclComm : SocketComm
{
    int device = 0;
    bool rccl_ok = false;
    std::vector<ncclComm_t> send_comm, recv_comm; // per peer: (me -> peer) and (peer -> me)
    std::vector<hipStream_t> send_stream, recv_stream;

    RcclComm(int rank_, int size_, const char *addr, int base_port) : SocketComm(rank_, size_, addr, base_port)
    {
        transport = PANGULU_AMD_TRANSPORT_RCCL;
        HIPC(hipGetDevice(&device));
        send_comm.assign((size_t)size, nullptr);
        recv_comm.assign((size_t)size, nullptr);
        send_stream.assign((size_t)size, nullptr);
        recv_stream.assign((size_t)size, nullptr);
        const char *to = getenv("PANGULU_AMD_RCCL_TIMEOUT_S");
        int timeout_s = to ? atoi(to) : 180;
        // initialise + self-test on a helper thread so a hang cannot take the run down with it
        auto fut = std::async(std::launch::async, [this]()
                              { return init_and_selftest(); });
        bool ok = false;
        if (fut.wait_for(std::chrono::seconds(timeout_s)) == std::future_status::ready)
            ok = fut.get();
        else
        {
            fprintf(stderr, "[PanguLU-AMD] rank %d: RCCL initialisation did not finish within %d s\n", rank, timeout_s);
            // the helper thread is abandoned; leak the future so its destructor does not join
            new std::future<bool>(std::move(fut));
        }
        // all ranks must agree (over TCP, which works regardless)
        i64 flag = ok ? 0 : 1;
        allreduce_sum_i64(&flag, 1);
        rccl_ok = flag == 0;
        if (!rccl_ok)
        {
            transport = PANGULU_AMD_TRANSPORT_HOST;
            if (rank == 0)
                fprintf(stderr, "[PanguLU-AMD] RCCL transport unavailable on %lld rank(s): falling back to host-staged block exchange\n", (long long)flag);
        }
    }

    bool init_and_selftest()
    {
        if (!R.load())
        {
            fprintf(stderr, "[PanguLU-AMD] rank %d: librccl.so not found\n", rank);
            return false;
        }
        if (hipSetDevice(device) != hipSuccess)
            return false;
        // communicators in one global order over ordered pairs (i, j): no rank can wait on a pair another rank
        // has not reached yet
        for (int i = 0; i < size; i++)
            for (int j = 0; j < size; j++)
            {
                if (i == j || (rank != i && rank != j))
                    continue;
                NcclUniqueId id;
                if (rank == i)
                {
                    if (R.GetUniqueId(&id) != 0)
                        return false;
                    send_bytes(j, TAG_NCCL_ID, &id, sizeof(id));
                    if (R.CommInitRank(&send_comm[(size_t)j], 2, id, 0) != 0)
                        return false;
                    if (hipStreamCreateWithFlags(&send_stream[(size_t)j], hipStreamNonBlocking) != hipSuccess)
                        return false;
                }
                else
                {
                    recv_bytes(i, TAG_NCCL_ID, &id, sizeof(id));
                    if (R.CommInitRank(&recv_comm[(size_t)i], 2, id, 1) != 0)
                        return false;
                    if (hipStreamCreateWithFlags(&recv_stream[(size_t)i], hipStreamNonBlocking) != hipSuccess)
                        return false;
                }
            }
        // self test: 1 MiB pattern over every directed pair, same global order
        const size_t N = 1 << 20;
        unsigned char *dbuf = nullptr;
        if (hipMalloc((void **)&dbuf, N) != hipSuccess)
            return false;
        std::vector<unsigned char> host(N);
        bool good = true;
        for (int i = 0; i < size && good; i++)
            for (int j = 0; j < size && good; j++)
            {
                if (i == j || (rank != i && rank != j))
                    continue;
                if (rank == i)
                {
                    for (size_t k = 0; k < N; k++)
                        host[k] = (unsigned char)((k * 131 + (size_t)i * 7 + (size_t)j) & 0xff);
                    good = hipMemcpy(dbuf, host.data(), N, hipMemcpyHostToDevice) == hipSuccess &&
                           R.Send(dbuf, N, NCCL_CHAR, 1, send_comm[(size_t)j], send_stream[(size_t)j]) == 0 &&
                           hipStreamSynchronize(send_stream[(size_t)j]) == hipSuccess;
                }
                else
                {
                    good = hipMemset(dbuf, 0, N) == hipSuccess &&
                           R.Recv(dbuf, N, NCCL_CHAR, 0, recv_comm[(size_t)i], recv_stream[(size_t)i]) == 0 &&
                           hipStreamSynchronize(recv_stream[(size_t)i]) == hipSuccess &&
                           hipMemcpy(host.data(), dbuf, N, hipMemcpyDeviceToHost) == hipSuccess;
                    for (size_t k = 0; k < N && good; k++)
                        good = host[k] == (unsigned char)((k * 131 + (size_t)i * 7 + (size_t)j) & 0xff);
                }
            }
        (void)hipFree(dbuf);
        return good;
    }

    ~RcclComm() override
    {
        if (rccl_ok)
        {
            for (int p = 0; p < size; p++)
            {
                if (send_stream[(size_t)p])
                    (void)hipStreamSynchronize(send_stream[(size_t)p]);
                if (send_comm[(size_t)p])
                    R.CommDestroy(send_comm[(size_t)p]);
                if (recv_comm[(size_t)p])
                    R.CommDestroy(recv_comm[(size_t)p]);
            }
        }
    }

    void isend_block(slot_t *s, const BlockHeader &h, int dst) override
    {
        if (!rccl_ok)
        {
            SocketComm::isend_block(s, h, dst);
            return;
        }
        // the caller has synchronised the compute stream: the record behind d_value is final
        size_t bytes = h.bytes_lo;
        post_announcement(dst, 1, h); // (never blocks; per-destination order = order of the ncclSends below)
        HIPC(hipSetDevice(device));
        NCCLC(R.Send((const char *)s->d_value - 32, bytes, NCCL_CHAR, 1, send_comm[(size_t)dst], send_stream[(size_t)dst]));
        sent_bytes += bytes;
    }

    void recv_block(slot_t *s, const BlockHeader &h, int src) override
    {
        if (!rccl_ok)
        {
            SocketComm::recv_block(s, h, src);
            return;
        }
        size_t bytes = h.bytes_lo;
        HIPC(hipSetDevice(device));
        NCCLC(R.Recv((char *)s->d_value - 32, bytes, NCCL_CHAR, 0, recv_comm[(size_t)src], recv_stream[(size_t)src]));
        // Waiting here is deadlock-free: the matching send was enqueued by the sender right after the announce we
        // just read, on a stream that only ever carries sends to this rank, in announce order.
        HIPC(hipStreamSynchronize(recv_stream[(size_t)src]));
        // the scheduler reads the pattern's nnz (colptr[nb]) and the header from the host mirror of the slot
        BlockHeader *rec = (BlockHeader *)((char *)s->value - 32);
        *rec = h;
        s->columnpointer[(((char *)s->rowindex - (char *)s->columnpointer) / sizeof(pangulu_inblock_ptr)) - 1] = (pangulu_inblock_ptr)h.nnz;
        recv_bytes_total += bytes;
    }

    void flush_sends() override
    {
        SocketComm::flush_sends();
        if (!rccl_ok)
            return;
        HIPC(hipSetDevice(device));
        for (int p = 0; p < size; p++)
            if (send_stream[(size_t)p])
                HIPC(hipStreamSynchronize(send_stream[(size_t)p]));
    }
};

} // namespace

Comm *make_rccl_comm(int rank, int size, const char *addr, int base_port, const void *nccl_id)
{
    (void)nccl_id; // per-pair ids are created and exchanged internally
    return new RcclComm(rank, size, addr, base_port);
}

int rccl_make_unique_id(void *out128)
{
    if (!R.load())
        return 1;
    NcclUniqueId id;
    if (R.GetUniqueId(&id) != 0)
        return 2;
    memcpy(out128, &id, sizeof(id));
    return 0;
}

} // namespace pg
