// pg_comm.cpp -- process group and block transport.
//
// Replaces the reference's MPI wrappers and point-to-point block exchange
// (src/pangulu_communication.c:3-105 rank/size/sync/bcast/isend/recv/probe, :1786-1944 recv_block/isend_block).
// The reference discovers source, identity and size of every block with MPI_Iprobe(ANY_SOURCE)
// (src/pangulu_numeric.c:33-35).  Device-to-device transports have no probe, so every transport here is split
// into a small TCP control plane on 127.0.0.1 (one ordered stream per rank pair: a 16-byte frame + the 32-byte
// record header announce each block) and a data plane:
//   HOST : the record follows its header on the same socket (host-staged, like the reference's MPI path);
//   RCCL : the record moves device-to-device with ncclSend/ncclRecv over xGMI (pg_comm_rccl.cpp).
// Sends are queued to a sender thread so the compute thread never blocks on a peer; payloads are final block
// records (never modified again), so they are sent in place.
#include "pg_comm_socket.h"

namespace pg
{

namespace
{

struct LoopbackComm : Comm
{
    void barrier() override {}
    void bcast(void *, size_t, int) override {}
    void allreduce_sum_i64(i64 *, int) override {}
    void allreduce_max_f64(double *, int) override {}
    void send_bytes(int, int, const void *, size_t) override { fatal("send on a single-rank group"); }
    void recv_bytes(int, int, void *, size_t) override { fatal("recv on a single-rank group"); }
    void isend_block(slot_t *, const BlockHeader &, int) override { fatal("block send on a single-rank group"); }
    bool probe_block(BlockHeader &, int &) override { return false; }
    void recv_block(slot_t *, const BlockHeader &, int) override {}
    void flush_sends() override {}
    bool set_send_gate(Marker *) override { return true; } // nothing is ever sent (PANGULU_AMD_FORCE_MULTI_LOOP on one rank)
};

Comm *g_world = nullptr;

} // namespace

Comm *world()
{
    if (!g_world)
        g_world = new LoopbackComm();
    return g_world;
}

// checker's build only (pangulu_amd_test_set_analysis_ranks): a one-process stand-in for a group of `size` ranks, good for
// PANGULU_AMD_ANALYSIS_ONLY handles -- the mapping and the models depend on the rank COUNT, not on any exchange
void set_fake_world(int size)
{
    set_world(nullptr);
    if (size > 1)
    {
        LoopbackComm *c = new LoopbackComm();
        c->size = size;
        g_world = c;
    }
}

void set_world(Comm *c)
{
    if (g_world && !g_world->abandoned)
        delete g_world;
    // (an abandoned object is leaked on purpose: its helper thread, stuck inside RCCL, still holds `this`)
    g_world = c;
}

Comm *make_socket_comm(int rank, int size, const char *addr, int base_port, int transport, const void *nccl_id)
{
    if (transport == PANGULU_AMD_TRANSPORT_RCCL)
        return make_rccl_comm(rank, size, addr, base_port, nccl_id);
    if (transport == PANGULU_AMD_TRANSPORT_IPC)
        return make_ipc_comm(rank, size, addr, base_port);
    SocketComm *c = new SocketComm(rank, size, addr, base_port);
    c->transport = PANGULU_AMD_TRANSPORT_HOST;
    return c;
}

} // namespace pg
