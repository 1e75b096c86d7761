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
// Start-up: the unique ids of all ordered pairs are exchanged over TCP by the calling thread; a helper thread then creates
// the communicators round by round of a round-robin tournament (disjoint pairs per round: 2 (N - 1) creations per rank,
// not N (N - 1) in sequence) and sends a test pattern over every directed pair.  If that does not finish within
// PANGULU_AMD_RCCL_TIMEOUT_S (default 90 s) on any rank, ALL ranks fall back together and say so; the abandoned
// helper only ever touches RCCL, never the control-plane sockets.
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

    std::vector<NcclUniqueId> id_out, id_in; // per peer: id of (me -> peer), made here; id of (peer -> me), made there
    double init_seconds = 0;

    RcclComm(int rank_, int size_, const char *addr, int base_port) : SocketComm(rank_, size_, addr, base_port)
    {
        transport = PANGULU_AMD_TRANSPORT_RCCL;
        const bool have_device = hipGetDevice(&device) == hipSuccess; // (a host-memory platform has none: fall back, do not fail)
        send_comm.assign((size_t)size, nullptr);
        recv_comm.assign((size_t)size, nullptr);
        send_stream.assign((size_t)size, nullptr);
        recv_stream.assign((size_t)size, nullptr);
        id_out.resize((size_t)size);
        id_in.resize((size_t)size);
        const char *to = getenv("PANGULU_AMD_RCCL_TIMEOUT_S");
        int timeout_s = to ? atoi(to) : 90;
        const double t0 = wall_seconds();
        // Everything that touches the TCP control plane happens HERE, on the calling thread, before the helper thread
        // exists: library check, agreement, the unique ids of all my ordered pairs.  The helper thread below only makes
        // RCCL calls, so abandoning it on a timeout cannot disturb the sockets or the mailbox the scheduler uses later.
        i64 missing = (have_device && R.load()) ? 0 : 1;
        if (missing)
            fprintf(stderr, "[PanguLU-AMD] rank %d: %s\n", rank, have_device ? "librccl.so not found" : "no HIP device for the RCCL transport");
        for (int p = 0; p < size && !missing; p++)
            if (p != rank && R.GetUniqueId(&id_out[(size_t)p]) != 0)
                missing = 1;
        allreduce_sum_i64(&missing, 1);
        bool ok = missing == 0;
        if (ok)
        {
            // (128-byte messages: they fit the socket buffers, so everybody can send first and receive afterwards)
            for (int p = 0; p < size; p++)
                if (p != rank)
                    send_bytes(p, TAG_NCCL_ID, &id_out[(size_t)p], sizeof(NcclUniqueId));
            for (int p = 0; p < size; p++)
                if (p != rank)
                    recv_bytes(p, TAG_NCCL_ID, &id_in[(size_t)p], sizeof(NcclUniqueId));
            // communicators + self-test on a helper thread so that a hang inside RCCL cannot take the run down with it
            auto fut = std::async(std::launch::async, [this]()
                                  { return init_and_selftest(); });
            if (fut.wait_for(std::chrono::seconds(timeout_s)) == std::future_status::ready)
                ok = fut.get();
            else
            {
                ok = false;
                fprintf(stderr, "[PanguLU-AMD] rank %d: RCCL initialisation did not finish within %d s\n", rank, timeout_s);
                // the helper thread is abandoned inside RCCL; leak the future so its destructor does not join, and this
                // object with it (set_world() checks `abandoned`): the thread still reads the ids and writes the communicator
                // tables if ncclCommInitRank ever returns
                new std::future<bool>(std::move(fut));
                abandoned = true;
            }
        }
        // all ranks must agree (over TCP, which works regardless)
        i64 flag = ok ? 0 : 1;
        allreduce_sum_i64(&flag, 1);
        rccl_ok = flag == 0;
        init_seconds = wall_seconds() - t0;
        if (!rccl_ok)
        {
            transport = PANGULU_AMD_TRANSPORT_HOST;
            if (rank == 0)
                fprintf(stderr, "[PanguLU-AMD] RCCL transport unavailable on %lld rank(s): falling back to host-staged block exchange\n", (long long)flag);
        }
        else if (rank == 0)
            fprintf(stderr, "[PanguLU-AMD] RCCL transport: %d ranks, %d two-rank communicators per rank, self-test passed, %.1f s\n", size,
                    2 * (size - 1), init_seconds);
    }

    // partner of `r` in round `round` of a round-robin tournament over m = size rounded up to even (circle method);
    // returns -1 when r sits the round out.  Every rank meets every other exactly once, and in one round all pairs are
    // disjoint -- so the communicators of a round are created concurrently and nobody waits for a rank that is busy
    // with somebody else.
    int partner_in_round(int r, int round) const
    {
        const int m = size + (size & 1);
        int p;
        if (r == m - 1)
            p = round;
        else if (r == round)
            p = m - 1;
        else
            p = ((2 * round - r) % (m - 1) + (m - 1)) % (m - 1);
        return (p >= size || p == r) ? -1 : p;
    }

    // RCCL calls only (see the constructor)
    bool init_and_selftest()
    {
        if (hipSetDevice(device) != hipSuccess)
            return false;
        const int m = size + (size & 1);
        for (int round = 0; round < m - 1; round++)
        {
            const int p = partner_in_round(rank, round);
            if (p < 0)
                continue;
            // both sides create (low -> high) first, then (high -> low): sender is rank 0 of the two-rank communicator
            for (int dir = 0; dir < 2; dir++)
            {
                const bool i_send = (dir == 0) == (rank < p);
                if (i_send)
                {
                    if (R.CommInitRank(&send_comm[(size_t)p], 2, id_out[(size_t)p], 0) != 0)
                        return false;
                    if (hipStreamCreateWithFlags(&send_stream[(size_t)p], hipStreamNonBlocking) != hipSuccess)
                        return false;
                }
                else
                {
                    if (R.CommInitRank(&recv_comm[(size_t)p], 2, id_in[(size_t)p], 1) != 0)
                        return false;
                    if (hipStreamCreateWithFlags(&recv_stream[(size_t)p], hipStreamNonBlocking) != hipSuccess)
                        return false;
                }
            }
        }
        // self test: a 1 MiB pattern over every directed pair, in the same tournament order
        const size_t N = 1 << 20;
        unsigned char *dbuf = nullptr;
        if (hipMalloc((void **)&dbuf, N) != hipSuccess)
            return false;
        std::vector<unsigned char> host(N);
        bool good = true;
        for (int round = 0; round < m - 1 && good; round++)
        {
            const int p = partner_in_round(rank, round);
            if (p < 0)
                continue;
            for (int dir = 0; dir < 2 && good; dir++)
            {
                const bool i_send = (dir == 0) == (rank < p);
                const int i = i_send ? rank : p, j = i_send ? p : rank;
                if (i_send)
                {
                    for (size_t k = 0; k < N; k++)
                        host[k] = (unsigned char)((k * 131 + (size_t)i * 7 + (size_t)j) & 0xff);
                    good = hipMemcpy(dbuf, host.data(), N, hipMemcpyHostToDevice) == hipSuccess &&
                           R.Send(dbuf, N, NCCL_CHAR, 1, send_comm[(size_t)p], send_stream[(size_t)p]) == 0 &&
                           hipStreamSynchronize(send_stream[(size_t)p]) == hipSuccess;
                }
                else
                {
                    good = hipMemset(dbuf, 0, N) == hipSuccess &&
                           R.Recv(dbuf, N, NCCL_CHAR, 0, recv_comm[(size_t)p], recv_stream[(size_t)p]) == 0 &&
                           hipStreamSynchronize(recv_stream[(size_t)p]) == hipSuccess &&
                           hipMemcpy(host.data(), dbuf, N, hipMemcpyDeviceToHost) == hipSuccess;
                    for (size_t k = 0; k < N && good; k++)
                        good = host[k] == (unsigned char)((k * 131 + (size_t)i * 7 + (size_t)j) & 0xff);
                }
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

    int rccl_ranks() const override { return rccl_ok ? size : 0; }

    // The ncclSend of a record is issued by the SENDER THREAD (SocketComm::sender_loop -> pass_gate), right before the
    // announcement is written: per destination the sends are issued in announcement order, this communicator's send side is
    // used by that one thread only, and the compute thread -- which posts under the scheduler's mutex -- never enters RCCL.
    // Gate: instead of waiting on the host for the marker behind the producing kernels, the send stream waits for it
    // (hipStreamWaitEvent: a Platform marker is a hipEvent_t on the back-end's stream): the transfer starts the moment the
    // record is final, no host thread in between.
    bool set_send_gate(Marker *marker) override
    {
        if (!rccl_ok)
            return false;
        send_gate = marker;
        return true;
    }

    void pass_gate(SendReq &r, void *ev) override
    {
        if (!r.dev_ptr)
        {
            SocketComm::pass_gate(r, ev);
            return;
        }
        HIPC(hipSetDevice(device));
        if (ev)
            HIPC(hipStreamWaitEvent(send_stream[(size_t)r.dst], (hipEvent_t)ev, 0));
        NCCLC(R.Send(r.dev_ptr, r.dev_bytes, NCCL_CHAR, 1, send_comm[(size_t)r.dst], send_stream[(size_t)r.dst]));
    }

    void isend_block(slot_t *s, const BlockHeader &h, int dst) override
    {
        if (!rccl_ok)
        {
            SocketComm::isend_block(s, h, dst);
            return;
        }
        // (without a gate the caller has synchronised the compute stream: the record behind d_value is final)
        const size_t bytes = h.bytes_lo;
        post_announcement(dst, 1, h, (const char *)s->d_value - 32, bytes); // (never blocks)
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
