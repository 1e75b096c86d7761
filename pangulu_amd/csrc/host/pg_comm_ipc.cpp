// pg_comm_ipc.cpp -- point-to-point block exchange by peer copies between the ranks' HBM arenas (one node, xGMI).
//
// The reference moves every block D2H -> MPI_Isend -> MPI_Recv -> H2D (…0201000.cu:196-201,
// src/pangulu_communication.c:1809,1850,1943).  On one node every rank's factor arena is a single hipMalloc
// allocation, finished blocks never move or change again, and HIP can map another process's allocation
// (hipIpcGetMemHandle / hipIpcOpenMemHandle, dmabuf based).  So a "send" is only an ANNOUNCEMENT on the TCP control
// plane of SocketComm -- frame + 32-byte record header whose reserved word carries the record's offset in the owner's
// arena -- and the consumer PULLS the record with one device-to-device copy (SDMA / blit over the direct xGMI link of
// the pair) into its receive slot.  No staging through host memory, no collective, no rendezvous between the two
// compute threads; traffic of different pairs shares nothing.
//
// The arena is a list of separately allocated chunks of at most 1 GiB (records never straddle one): mapping a peer
// allocation above 2 GiB was seen to block forever inside hipIpcOpenMemHandle on this stack.
// register_arena() is collective: handles are exchanged over TCP, every peer arena is opened and a test read of each
// is compared with bytes the owner sent over TCP.  If any rank fails any step, ALL ranks fall back to the host-staged
// path together and say so.  Records outside the registered arena (there are none today) are sent host-staged, block
// by block; the receiver tells the two apart by the header's reserved word.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "pg_comm_socket.h"

namespace pg
{

namespace
{

enum : int
{
    TAG_IPC_HANDLE = 0x7ffffe00,
    TAG_IPC_PROBE = 0x7ffffe01
};

struct IpcComm : SocketComm
{
    int device = 0;
    bool ipc_ok = false;
    std::vector<char *> my_chunks;
    size_t my_chunk_bytes = 0, my_total = 0;
    std::vector<std::vector<char *>> peer_chunks; // [rank][chunk]
    std::vector<size_t> peer_chunk_bytes;
    hipStream_t copy_stream = nullptr;

    IpcComm(int rank_, int size_, const char *addr, int base_port) : SocketComm(rank_, size_, addr, base_port)
    {
        transport = PANGULU_AMD_TRANSPORT_IPC;
        peer_chunks.assign((size_t)size, {});
        peer_chunk_bytes.assign((size_t)size, 0);
        if (hipGetDevice(&device) != hipSuccess)
            device = 0;
    }

    ~IpcComm() override { close_peers(); }

    void close_peers()
    {
        for (auto &v : peer_chunks)
        {
            for (char *p : v)
                if (p)
                    (void)hipIpcCloseMemHandle(p);
            v.clear();
        }
        if (copy_stream)
        {
            (void)hipStreamDestroy(copy_stream);
            copy_stream = nullptr;
        }
    }

    struct Hello
    {
        int usable;
        unsigned nchunks;
        unsigned long long chunk_bytes, total_bytes;
        unsigned char probe[256]; // first bytes of chunk 0, for the test read
    };

    // collective over all ranks; nchunks == 0 (a host-memory platform) votes for the fallback
    void register_arena(char *const *chunks, size_t nchunks, size_t chunk_bytes, size_t total_bytes) override
    {
        close_peers();
        ipc_ok = false;
        if (nchunks > 0 && hipGetDevice(&device) != hipSuccess) // (the caller has selected this rank's device by now)
            device = 0;
        const bool trace = getenv("PANGULU_AMD_TRACE") != nullptr;
        if (trace)
            fprintf(stderr, "[pangulu_amd trace] rank %d: registering a %.2f GB arena (%zu chunks) for peer copies\n", rank, (double)total_bytes / 1e9, nchunks);
        my_chunks.assign(chunks, chunks + nchunks);
        my_chunk_bytes = chunk_bytes;
        my_total = total_bytes;
        bool good = nchunks > 0 && total_bytes >= 256;
        std::vector<hipIpcMemHandle_t> mine(nchunks);
        if (good)
            good = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) == hipSuccess;
        for (size_t c = 0; c < nchunks && good; c++)
            good = hipIpcGetMemHandle(&mine[c], chunks[c]) == hipSuccess;
        Hello hello;
        memset(&hello, 0, sizeof(hello));
        hello.nchunks = (unsigned)nchunks;
        hello.chunk_bytes = chunk_bytes;
        hello.total_bytes = total_bytes;
        if (good && hipMemcpy(hello.probe, chunks[0], sizeof(hello.probe), hipMemcpyDeviceToHost) != hipSuccess)
            good = false;
        if (!good)
            (void)hipGetLastError();
        hello.usable = good ? 1 : 0;
        // everyone tells everyone: the hello, then (if usable) the handles of all chunks
        for (int r = 0; r < size; r++)
            if (r != rank)
            {
                send_bytes(r, TAG_IPC_HANDLE, &hello, sizeof(hello));
                if (hello.usable)
                    send_bytes(r, TAG_IPC_PROBE, mine.data(), sizeof(hipIpcMemHandle_t) * nchunks);
            }
        std::vector<Hello> all((size_t)size);
        std::vector<std::vector<hipIpcMemHandle_t>> handles((size_t)size);
        for (int r = 0; r < size; r++)
        {
            if (r == rank)
                continue;
            recv_bytes(r, TAG_IPC_HANDLE, &all[(size_t)r], sizeof(Hello));
            if (!all[(size_t)r].usable)
            {
                good = false;
                continue;
            }
            handles[(size_t)r].resize(all[(size_t)r].nchunks);
            recv_bytes(r, TAG_IPC_PROBE, handles[(size_t)r].data(), sizeof(hipIpcMemHandle_t) * all[(size_t)r].nchunks);
        }
        // One exporter at a time: while the others map rank e's chunks, rank e waits in a TCP barrier and makes no
        // HIP call.
        for (int e = 0; e < size; e++)
        {
            if (e != rank && good)
            {
                peer_chunk_bytes[(size_t)e] = (size_t)all[(size_t)e].chunk_bytes;
                for (unsigned c = 0; c < all[(size_t)e].nchunks && good; c++)
                {
                    void *p = nullptr;
                    if (hipIpcOpenMemHandle(&p, handles[(size_t)e][c], hipIpcMemLazyEnablePeerAccess) != hipSuccess)
                    {
                        (void)hipGetLastError();
                        fprintf(stderr, "[PanguLU-AMD] rank %d: cannot map arena chunk %u of rank %d\n", rank, c, e);
                        good = false;
                        break;
                    }
                    peer_chunks[(size_t)e].push_back((char *)p);
                }
                if (good)
                {
                    if (trace)
                        fprintf(stderr, "[pangulu_amd trace] rank %d: mapped the arena of rank %d\n", rank, e);
                    // test read through the mapping
                    unsigned char *dtmp = nullptr, back[256];
                    bool ok = hipMalloc((void **)&dtmp, sizeof(back)) == hipSuccess &&
                              hipMemcpyAsync(dtmp, peer_chunks[(size_t)e][0], sizeof(back), hipMemcpyDeviceToDevice, copy_stream) == hipSuccess &&
                              hipStreamSynchronize(copy_stream) == hipSuccess &&
                              hipMemcpy(back, dtmp, sizeof(back), hipMemcpyDeviceToHost) == hipSuccess &&
                              memcmp(back, all[(size_t)e].probe, sizeof(back)) == 0;
                    if (dtmp)
                        (void)hipFree(dtmp);
                    if (!ok)
                    {
                        (void)hipGetLastError();
                        fprintf(stderr, "[PanguLU-AMD] rank %d: test read from the arena of rank %d failed\n", rank, e);
                        good = false;
                    }
                }
            }
            barrier();
        }
        i64 bad = good ? 0 : 1;
        allreduce_sum_i64(&bad, 1);
        ipc_ok = bad == 0;
        if (!ipc_ok)
        {
            close_peers();
            transport = PANGULU_AMD_TRANSPORT_HOST;
            if (rank == 0)
                fprintf(stderr, "[PanguLU-AMD] peer-copy transport unavailable on %lld rank(s): falling back to host-staged block exchange\n", (long long)bad);
        }
        else
            transport = PANGULU_AMD_TRANSPORT_IPC;
        if (trace)
            fprintf(stderr, "[pangulu_amd trace] rank %d: peer copies %s\n", rank, ipc_ok ? "enabled" : "disabled");
    }

    // offset of a device record in this rank's arena, or ~0 when it lies outside
    unsigned long long arena_offset(const char *rec, size_t bytes) const
    {
        for (size_t c = 0; c < my_chunks.size(); c++)
        {
            const char *b = my_chunks[c];
            const size_t len = std::min(my_chunk_bytes, my_total - c * my_chunk_bytes);
            if (rec >= b && rec + bytes <= b + len)
                return (unsigned long long)c * my_chunk_bytes + (unsigned long long)(rec - b);
        }
        return ~0ull;
    }

    void isend_block(slot_t *s, const BlockHeader &h, int dst) override
    {
        const char *rec = (const char *)s->d_value - 32;
        const unsigned long long off = ipc_ok ? arena_offset(rec, h.bytes_lo) : ~0ull;
        if (off == ~0ull)
        {
            BlockHeader plain = h;
            plain.reserved = 0; // payload follows on the socket
            SocketComm::isend_block(s, plain, dst);
            return;
        }
        // the caller has synchronised the compute stream: the record behind d_value is final and stays where it is
        BlockHeader ann = h;
        ann.reserved = off + 1; // (+1: zero means "payload follows")
        post_announcement(dst, 2, ann);
        sent_bytes += h.bytes_lo;
    }

    // announcements can wait for a marker in the sender thread (the data itself never moves on this side)
    bool set_send_gate(Marker *marker) override
    {
        if (!ipc_ok || !active_platform().marker_wait)
            return false;
        send_gate = marker;
        return true;
    }

    void recv_block(slot_t *s, const BlockHeader &h, int src) override
    {
        recv_block_begin(s, h, src);
        recv_blocks_finish();
    }

    // the pull is only queued here: the receive thread announces a whole burst of blocks to the copy engine and waits
    // once (a synchronous copy per block costs 20-30 us of API latency each, tens of thousands of times)
    void recv_block_begin(slot_t *s, const BlockHeader &h, int src) override
    {
        if (h.reserved == 0)
        {
            SocketComm::recv_block(s, h, src);
            return;
        }
        const unsigned long long off = h.reserved - 1;
        const size_t cb = peer_chunk_bytes[(size_t)src];
        if (!ipc_ok || cb == 0 || off / cb >= peer_chunks[(size_t)src].size())
            fatal("rank %d: peer-copy announcement from rank %d outside its mapped arena", rank, src);
        const size_t bytes = h.bytes_lo;
        const char *from = peer_chunks[(size_t)src][off / cb] + off % cb;
        if (hipSetDevice(device) != hipSuccess ||
            hipMemcpyAsync((char *)s->d_value - 32, from, bytes, hipMemcpyDeviceToDevice, copy_stream) != hipSuccess)
            fatal("rank %d: peer copy of block (%u,%u) from rank %d failed: %s", rank, h.brow, h.bcol, src, hipGetErrorString(hipGetLastError()));
        copies_in_flight = true;
        // the scheduler reads the pattern's nnz (colptr[nb]) and the header from the host mirror of the slot
        BlockHeader *rec = (BlockHeader *)((char *)s->value - 32);
        *rec = h;
        rec->reserved = 0;
        s->columnpointer[(((char *)s->rowindex - (char *)s->columnpointer) / sizeof(pangulu_inblock_ptr)) - 1] = (pangulu_inblock_ptr)h.nnz;
        recv_bytes_total += bytes;
    }

    void recv_blocks_finish() override
    {
        if (!copies_in_flight)
            return;
        if (hipStreamSynchronize(copy_stream) != hipSuccess)
            fatal("rank %d: peer copies failed: %s", rank, hipGetErrorString(hipGetLastError()));
        copies_in_flight = false;
    }
    bool copies_in_flight = false;
};

} // namespace

Comm *make_ipc_comm(int rank, int size, const char *addr, int base_port)
{
    return new IpcComm(rank, size, addr, base_port);
}

} // namespace pg
