// TEST INFRASTRUCTURE, not product code: an in-process stand-in for the eight RCCL entry points that tpnet_amd/csrc/rows_rccl.hip
// resolves with dlsym, so that the `comm && G > 1` branch of tpnet_rows_step_targeted (message offsets, halo placement, two
// messages per peer inside one group) executes on a ONE-GPU box.  The "ranks" are host THREADS of one process, each with its own
// HIP stream and its own communicator handle, each driving its shard exactly as a rank of a real job does.  Semantics kept from
// NCCL: the operations of a group are issued at ncclGroupEnd; a send and a receive between the same two ranks match in the order
// they were posted; on the receiver's stream the data is there when the receive has completed; on the sender's stream the buffer
// may be rewritten by whatever follows the send.  How:
//   * at its group end a rank records an event on its stream ("my send buffers are packed") and publishes its sends;
//   * for each of its receives it waits (host side, condition variable) for the matching send to be published, then enqueues on
//     its OWN stream: wait for the sender's event, copy; it records an event behind the copy and hands it to the sender;
//   * for each of its sends it waits (host side) until the receiver has enqueued the copy, and makes its own stream wait for
//     the copy's event.
// Every dependency is an event that was recorded before it is waited for: nothing spins on the device.  Wire time is not
// modelled; nothing here measures anything.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <vector>

namespace {

struct UniqueId { char internal[128]; };
struct Group;

struct Op {
    bool is_send;
    int peer;
    const void* sptr;
    void* rptr;
    size_t bytes;
};

struct Comm {
    Group* g;
    int rank;
    std::vector<Op> pending;        // this rank's operations of the open group
    hipStream_t stream = nullptr;
};

struct Posted {                     // a published send
    int from, to;
    const void* sptr;
    size_t bytes;
    hipEvent_t packed;              // recorded on the sender's stream at its group end
    bool consumed = false;
    hipEvent_t copied = nullptr;    // recorded on the receiver's stream behind the copy
    unsigned long long seq;
};

struct Group {
    char id[128];
    int nranks;
    std::vector<Comm*> comms;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Posted> sends;       // published, not yet retired by their sender
    unsigned long long next_seq = 0;
};

std::mutex g_mu;
std::vector<Group*> g_groups;
uint32_t g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Comm*> t_open;        // communicators with operations in this thread's open group
std::atomic<long> g_counts[4];                 // sends, receives, copies, host waits that timed out
constexpr int kWaitSeconds = 20;

size_t type_bytes(int t) { return t == 7 ? 4 : (t == 8 ? 8 : (t == 0 || t == 1) ? 1 : 4); }   // ncclFloat32 = 7, ncclFloat64 = 8

int flush(Comm* c) {
    Group* g = c->g;
    hipStream_t s = c->stream;
    hipEvent_t packed;
    if (hipEventCreateWithFlags(&packed, hipEventDisableTiming) != hipSuccess || hipEventRecord(packed, s) != hipSuccess) return 1;
    std::vector<unsigned long long> mine;
    std::unique_lock<std::mutex> lk(g->mu);
    for (const Op& op : c->pending) {
        if (!op.is_send) continue;
        Posted p;
        p.from = c->rank; p.to = op.peer; p.sptr = op.sptr; p.bytes = op.bytes; p.packed = packed; p.seq = g->next_seq++;
        g->sends.push_back(p);
        mine.push_back(p.seq);
    }
    g->cv.notify_all();
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(kWaitSeconds);
    for (const Op& op : c->pending) {
        if (op.is_send) continue;
        Posted* hit = nullptr;
        for (;;) {
            for (Posted& p : g->sends)
                if (!p.consumed && p.from == op.peer && p.to == c->rank) { hit = &p; break; }   // the oldest one: posting order
            if (hit) break;
            if (g->cv.wait_until(lk, deadline) == std::cv_status::timeout) { ++g_counts[3]; return 1; }
        }
        if (hit->bytes != op.bytes) return 1;
        if (hipStreamWaitEvent(s, hit->packed, 0) != hipSuccess) return 1;
        if (hipMemcpyAsync(op.rptr, hit->sptr, op.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
        if (hipEventCreateWithFlags(&hit->copied, hipEventDisableTiming) != hipSuccess || hipEventRecord(hit->copied, s) != hipSuccess) return 1;
        hit->consumed = true;
        ++g_counts[2];
        g->cv.notify_all();
    }
    for (unsigned long long seq : mine) {
        for (;;) {
            Posted* p = nullptr;
            for (Posted& q : g->sends)
                if (q.seq == seq) p = &q;
            if (!p) return 1;
            if (p->consumed) {
                if (hipStreamWaitEvent(s, p->copied, 0) != hipSuccess) return 1;
                break;
            }
            if (g->cv.wait_until(lk, deadline) == std::cv_status::timeout) { ++g_counts[3]; return 1; }
        }
    }
    // retire my consumed sends (the events stay alive: a stream may still wait for them)
    for (auto it = g->sends.begin(); it != g->sends.end();)
        it = (it->from == c->rank && it->consumed) ? g->sends.erase(it) : it + 1;
    c->pending.clear();
    return 0;
}

int post(Comm* c, Op op, hipStream_t s) {
    if (c->pending.empty()) c->stream = s;
    else if (c->stream != s) return 1;              // (one stream per group, as the product uses it)
    c->pending.push_back(op);
    bool known = false;
    for (Comm* x : t_open) known |= (x == c);
    if (!known) t_open.push_back(c);
    if (t_depth == 0) {                             // outside a group: a group of one
        t_open.clear();
        return flush(c);
    }
    return 0;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void* id) {
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0, 128);
    memcpy(id, &g_next_id, sizeof(g_next_id));
    memcpy((char*)id + 8, "tpnet-loopback", 14);
    ++g_next_id;
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return 1;
    std::lock_guard<std::mutex> lk(g_mu);
    Group* g = nullptr;
    for (Group* x : g_groups)
        if (memcmp(x->id, id.internal, 128) == 0) g = x;
    if (!g) {
        g = new Group();
        memcpy(g->id, id.internal, 128);
        g->nranks = nranks;
        g->comms.assign((size_t)nranks, nullptr);
        g_groups.push_back(g);
    }
    if (g->nranks != nranks || g->comms[(size_t)rank]) return 1;
    Comm* c = new Comm();
    c->g = g;
    c->rank = rank;
    g->comms[(size_t)rank] = c;
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    c->g->comms[(size_t)c->rank] = nullptr;
    delete c;
    return 0;
}

int ncclGroupStart() { ++t_depth; return 0; }

int ncclGroupEnd() {
    if (t_depth > 0) --t_depth;
    int bad = 0;
    if (t_depth == 0) {
        std::vector<Comm*> open;
        open.swap(t_open);
        for (Comm* c : open) bad |= flush(c);
    }
    return bad;
}

int ncclAllGather(const void*, void*, size_t, int, void*, hipStream_t) { return 1; }     // (not what this stand-in is for)

int ncclSend(const void* buf, size_t count, int type, int peer, void* comm, hipStream_t s) {
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->g->nranks || peer == c->rank) return 1;
    ++g_counts[0];
    return post(c, Op{true, peer, buf, nullptr, count * type_bytes(type)}, s);
}

int ncclRecv(void* buf, size_t count, int type, int peer, void* comm, hipStream_t s) {
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->g->nranks || peer == c->rank) return 1;
    ++g_counts[1];
    return post(c, Op{false, peer, nullptr, buf, count * type_bytes(type)}, s);
}

// ---- for the test: what happened
long tpnet_loopback_count(int which) { return (which >= 0 && which < 4) ? g_counts[which].load() : -1; }
long tpnet_loopback_pending(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    long n = 0;
    for (Group* g : g_groups) {
        std::lock_guard<std::mutex> lk2(g->mu);
        n += (long)g->sends.size();
        for (Comm* c : g->comms)
            if (c) n += (long)c->pending.size();
    }
    return n;
}

}  // extern "C"
