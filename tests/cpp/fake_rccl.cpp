// fake_rccl.cpp -> libfake_rccl.so: a TEST DOUBLE for librccl.  Test infrastructure only: never loaded by the package or
// by bench.py; the tests point voidin_amd/csrc/dist.hip at it through $VD_RCCL_LIB so that the `world > 1` code of the
// C-ABI exchange (vd_dist_step_full_dev / _draws_dev / _indices_dev: the offset arithmetic, the unequal last shard, ranks
// with n_local == 0, the grouped ncclSend / ncclRecv block) runs on ONE GPU.  Real RCCL refuses two ranks per device.
//
// It exports exactly the ten symbols dist.hip resolves (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather,
// ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString, ncclGetVersion) with RCCL's semantics as far
// as a caller on one stream can observe them:
//   * ranks are PROCESSES that share a device; ncclCommInitRank is a rendezvous through a file-backed shared segment
//     (mmap MAP_SHARED under $VD_FAKE_RCCL_DIR, default /dev/shm) keyed by the 128-byte id;
//   * a collective / a group of point-to-point operations completes in stream order: the call synchronises the stream,
//     copies its send buffers device -> its shared "outbox", posts one descriptor per destination, then lands every
//     expected message from the peers' outboxes host -> device.  Later work on the stream sees the received data, as
//     it would behind a real stream-ordered collective (graph capture is not supported - the call blocks);
//   * sends and receives between a pair match in posting order and their sizes must agree (a mismatch is
//     ncclInvalidArgument here, where RCCL would hang or truncate).  Every exchange (a collective, or a group with
//     operations) carries the communicator's running number - all ranks issue the same sequence of them, as dist.hip
//     does - so a receive never takes a message of a LATER exchange: if the peer has moved on without sending (its send
//     failed), the receive reports ncclSystemError at once and the later message stays for the exchange it belongs to;
//   * every wait is bounded ($VD_FAKE_RCCL_TIMEOUT_S, default 60): a missing peer is ncclSystemError, never a hang.
// Extras for the tests: vd_fake_rccl_stats (what actually ran) and $VD_FAKE_RCCL_FAIL_SEND_AT=k (the k-th ncclSend of
// the process fails, the error path of the caller).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr int kMaxRanks = 16;
constexpr int kRing = 8;   // messages a (src, dst) pair may have in flight = sends to one peer per group

struct Msg { uint64_t offset, bytes, seq; };
struct Channel {
    std::atomic<uint64_t> posted, consumed;
    Msg ring[kRing];
};
struct Header {
    std::atomic<uint32_t> arrived, departed;
    uint32_t nranks, pad;
    uint64_t outbox_bytes;
    Channel ch[kMaxRanks][kMaxRanks];   // [src][dst]
};

struct Op { bool send; void* buf; uint64_t bytes; int peer; };

}  // namespace

struct ncclComm {   // the opaque ncclComm_t of rccl.h points here
    int rank = 0, world = 1;
    Header* hdr = nullptr;
    char* outbox[kMaxRanks] = {nullptr};
    uint64_t outbox_bytes = 0;
    uint64_t seq = 0;                   // exchanges issued on this communicator by this rank
    std::string dir, key;
};

namespace {

struct Stats { uint64_t allgathers, groups, sends, recvs, bytes_sent, bytes_received, failed_sends; };
Stats g_stats = {};
uint64_t g_send_calls = 0;

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local ncclComm* t_comm = nullptr;
thread_local hipStream_t t_stream = nullptr;
thread_local ncclResult_t t_group_err = ncclSuccess;

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double timeout_s() {
    const char* e = getenv("VD_FAKE_RCCL_TIMEOUT_S");
    return e && *e ? atof(e) : 60.0;
}
template <typename F>
bool wait_until(F cond) {
    const double t0 = now_s(), lim = timeout_s();
    for (unsigned spin = 0; !cond(); ++spin) {
        if (spin > 64) usleep(50);
        if ((spin & 1023u) == 1023u && now_s() - t0 > lim) return false;
    }
    return true;
}

void* map_file(const std::string& path, uint64_t bytes, bool create) {
    const int fd = open(path.c_str(), create ? (O_RDWR | O_CREAT) : O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); return nullptr; }   // new pages read as zero
    struct stat st;
    if (fstat(fd, &st) != 0 || (uint64_t)st.st_size < bytes) { close(fd); return nullptr; }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    return p == MAP_FAILED ? nullptr : p;
}

size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

// One stream-ordered exchange: all sends of `ops` are posted, then all receives are landed.
ncclResult_t run_ops(ncclComm* c, const std::vector<Op>& ops, hipStream_t stream) {
    if (ops.empty()) return ncclSuccess;
    const uint64_t seq = ++c->seq;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    Header* h = c->hdr;
    const int me = c->rank;
    // the outbox is reused: every message of the previous exchange must have been taken
    for (int q = 0; q < c->world; ++q) {
        Channel& ch = h->ch[me][q];
        if (!wait_until([&] { return ch.consumed.load(std::memory_order_acquire) == ch.posted.load(std::memory_order_relaxed); }))
            return ncclSystemError;
    }
    struct Staged { void* buf; uint64_t bytes, offset; };
    std::vector<Staged> staged;   // one copy per distinct (buffer, size): an all-gather sends the same bytes to every peer
    uint64_t fill = 0;
    int per_peer[kMaxRanks] = {0};
    for (const Op& op : ops) {
        if (!op.send) continue;
        if (op.peer < 0 || op.peer >= c->world || op.peer == me) return ncclInvalidArgument;
        if (++per_peer[op.peer] > kRing) return ncclInvalidUsage;
        uint64_t off = UINT64_MAX;
        for (const Staged& s : staged) if (s.buf == op.buf && s.bytes == op.bytes) off = s.offset;
        if (off == UINT64_MAX) {
            off = fill;
            fill += (op.bytes + 63u) & ~(uint64_t)63u;
            if (fill > c->outbox_bytes) return ncclInternalError;   // raise VD_FAKE_RCCL_OUTBOX_MB
            if (op.bytes && hipMemcpy(c->outbox[me] + off, op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            staged.push_back({op.buf, op.bytes, off});
        }
        Channel& ch = h->ch[me][op.peer];
        const uint64_t k = ch.posted.load(std::memory_order_relaxed);
        ch.ring[k % kRing] = {off, op.bytes, seq};
        ch.posted.store(k + 1, std::memory_order_release);
        g_stats.sends++; g_stats.bytes_sent += op.bytes;
    }
    ncclResult_t first = ncclSuccess;   // a failed receive does not stop the others: the channels stay in step
    for (const Op& op : ops) {
        if (op.send) continue;
        if (op.peer < 0 || op.peer >= c->world || op.peer == me) { if (first == ncclSuccess) first = ncclInvalidArgument; continue; }
        Channel& ch = h->ch[op.peer][me];
        uint64_t k = ch.consumed.load(std::memory_order_relaxed);
        Msg m = {0, 0, 0};
        bool have = false;
        for (;;) {
            if (!wait_until([&] { return ch.posted.load(std::memory_order_acquire) > k; })) break;
            m = ch.ring[k % kRing];
            if (m.seq == seq) { have = true; break; }
            if (m.seq > seq) break;                               // the peer is past this exchange: nothing will come
            ch.consumed.store(++k, std::memory_order_release);    // left over from an exchange this rank gave up on
        }
        if (!have) { if (first == ncclSuccess) first = ncclSystemError; continue; }
        ncclResult_t r = ncclSuccess;
        if (m.bytes != op.bytes) r = ncclInvalidArgument;
        else if (m.bytes && hipMemcpy(op.buf, c->outbox[op.peer] + m.offset, m.bytes, hipMemcpyHostToDevice) != hipSuccess) r = ncclUnhandledCudaError;
        ch.consumed.store(k + 1, std::memory_order_release);   // taken either way: the sender must not wait for ever
        if (r != ncclSuccess) { if (first == ncclSuccess) first = r; continue; }
        g_stats.recvs++; g_stats.bytes_received += op.bytes;
    }
    return first;
}

ncclResult_t submit(ncclComm* c, hipStream_t stream, const Op* ops, size_t n) {
    if (!c) return ncclInvalidArgument;
    if (t_depth > 0) {
        if (t_comm && (t_comm != c || t_stream != stream)) return ncclInvalidUsage;   // one communicator and stream per group
        t_comm = c; t_stream = stream;
        t_ops.insert(t_ops.end(), ops, ops + n);
        return ncclSuccess;
    }
    return run_ops(c, std::vector<Op>(ops, ops + n), stream);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* version) {
    if (!version) return ncclInvalidArgument;
    *version = 99901;   // not a version RCCL ever had: VdDistInfo.rccl_version shows that the double was bound
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "fake rccl: no error";
        case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
        case ncclSystemError: return "fake rccl: system error (peer missing / timed out, or an injected failure)";
        case ncclInternalError: return "fake rccl: internal error (outbox too small?)";
        case ncclInvalidArgument: return "fake rccl: invalid argument (or mismatched send / recv sizes)";
        case ncclInvalidUsage: return "fake rccl: invalid usage";
        default: return "fake rccl: error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    unsigned char rnd[16] = {0};
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) { (void)!fread(rnd, 1, sizeof(rnd), f); fclose(f); }
    snprintf(id->internal, sizeof(id->internal), "vdfake-%d-", (int)getpid());
    const size_t l = strlen(id->internal);
    for (size_t i = 0; i < sizeof(rnd) && l + 2 * i + 2 < sizeof(id->internal); ++i) snprintf(id->internal + l + 2 * i, 3, "%02x", rnd[i]);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->world = nranks;
    const char* dir = getenv("VD_FAKE_RCCL_DIR");
    c->dir = dir && *dir ? dir : "/dev/shm";
    id.internal[sizeof(id.internal) - 1] = 0;
    c->key = id.internal;
    for (char& ch : c->key) if (!((ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || ch == '-')) ch = '_';
    const char* mb = getenv("VD_FAKE_RCCL_OUTBOX_MB");
    c->outbox_bytes = (uint64_t)(mb && *mb ? atoll(mb) : 512) << 20;   // sparse until written
    c->hdr = static_cast<Header*>(map_file(c->dir + "/" + c->key + ".hdr", sizeof(Header), true));
    c->outbox[rank] = static_cast<char*>(map_file(c->dir + "/" + c->key + ".out" + std::to_string(rank), c->outbox_bytes, true));
    if (!c->hdr || !c->outbox[rank]) { delete c; return ncclSystemError; }
    c->hdr->nranks = (uint32_t)nranks;
    c->hdr->outbox_bytes = c->outbox_bytes;
    c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel);
    if (!wait_until([&] { return c->hdr->arrived.load(std::memory_order_acquire) >= (uint32_t)nranks; })) { delete c; return ncclSystemError; }
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) continue;
        c->outbox[q] = static_cast<char*>(map_file(c->dir + "/" + c->key + ".out" + std::to_string(q), c->outbox_bytes, false));
        if (!c->outbox[q]) { delete c; return ncclSystemError; }
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    // peers may still be reading this rank's outbox: leave only when everything posted was taken (bounded)
    for (int q = 0; q < c->world; ++q) {
        Channel& ch = c->hdr->ch[c->rank][q];
        (void)wait_until([&] { return ch.consumed.load(std::memory_order_acquire) == ch.posted.load(std::memory_order_relaxed); });
    }
    const uint32_t gone = c->hdr->departed.fetch_add(1, std::memory_order_acq_rel) + 1;
    unlink((c->dir + "/" + c->key + ".out" + std::to_string(c->rank)).c_str());
    if (gone == (uint32_t)c->world) unlink((c->dir + "/" + c->key + ".hdr").c_str());
    for (int q = 0; q < c->world; ++q) if (c->outbox[q]) munmap(c->outbox[q], c->outbox_bytes);
    munmap(c->hdr, sizeof(Header));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    if (t_depth++ == 0) { t_ops.clear(); t_comm = nullptr; t_stream = nullptr; t_group_err = ncclSuccess; }
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    g_stats.groups++;
    ncclResult_t r = t_comm ? run_ops(t_comm, t_ops, t_stream) : ncclSuccess;
    t_ops.clear(); t_comm = nullptr;
    return r;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t stream) {
    const char* fail = getenv("VD_FAKE_RCCL_FAIL_SEND_AT");
    if (fail && *fail && ++g_send_calls == (uint64_t)atoll(fail)) { g_stats.failed_sends++; return ncclSystemError; }
    const Op op = {true, const_cast<void*>(buf), (uint64_t)count * dtype_bytes(dt), peer};
    return submit(c, stream, &op, 1);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t stream) {
    const Op op = {false, buf, (uint64_t)count * dtype_bytes(dt), peer};
    return submit(c, stream, &op, 1);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t stream) {
    if (!c || !send || !recv) return ncclInvalidArgument;
    const uint64_t bytes = (uint64_t)count * dtype_bytes(dt);
    g_stats.allgathers++;
    char* own = static_cast<char*>(recv) + (uint64_t)c->rank * bytes;
    if (own != send && bytes && hipMemcpyAsync(own, send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<Op> ops;
    for (int q = 0; q < c->world; ++q) if (q != c->rank) ops.push_back({true, const_cast<void*>(send), bytes, q});
    for (int q = 0; q < c->world; ++q) if (q != c->rank) ops.push_back({false, static_cast<char*>(recv) + (uint64_t)q * bytes, bytes, q});
    return submit(c, stream, ops.data(), ops.size());
}

// what ran in this process (the tests assert that the world > 1 exchange really went through here)
void vd_fake_rccl_stats(uint64_t out[7]) {
    out[0] = g_stats.allgathers; out[1] = g_stats.groups; out[2] = g_stats.sends; out[3] = g_stats.recvs;
    out[4] = g_stats.bytes_sent; out[5] = g_stats.bytes_received; out[6] = g_stats.failed_sends;
}

}  // extern "C"
