// shm_gather.hpp -- all-gather of small per-rank blocks between the processes of ONE node through a POSIX
// shared-memory segment (no HIP in this file: the engine's results already land in host memory, which is where the
// outer-loop algebra reads them, src/muse.jl:177-188).
//
// The reference gathers the results of its map to the master process over Distributed's sockets
// (src/util.jl:74-83).  With one process per GPU on one node the per-rank score blocks are a few kilobytes that
// every rank's HOST needs: sending them GPU -> GPU over xGMI with a collective kernel and then down to every host
// (the RCCL transport, muse_comm.cpp) costs 35-40 us per step, all of it latency.  Here every rank copies its block
// from its own pinned result area into its slot of the segment and publishes a sequence number; readers wait for
// the numbers of all ranks.  Cost: one cache-line handoff per rank.
//
// Layout:  Header (one page) | lines[narea][nranks] (64 B each: ready, consumed) | blocks[narea][nranks][block_doubles]
// Protocol per (area, sequence number s = 1, 2, ...), the same call sequence on every rank:
//   writer r:  wait until consumed[area][q] >= s-1 for every q   (everyone has copied my previous block out)
//              fill blocks[area][r]; ready[area][r] = s (release)
//   reader r:  wait until ready[area][q] >= s for every q (acquire); copy; consumed[area][r] = s (release)
// Every wait is bounded (timeout) and gives up at once when any rank has raised the segment's abort word.
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <string>

#ifndef MUSE_CPU_RELAX  // a polite spin on every host the library builds on
#if defined(__x86_64__) || defined(__i386__)
#define MUSE_CPU_RELAX() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define MUSE_CPU_RELAX() asm volatile("yield" ::: "memory")
#else
#define MUSE_CPU_RELAX() sched_yield()
#endif
#endif

namespace muse_shm {

constexpr uint64_t kMagic = 0x314d48534553554dull;  // "MUSESHM1"
constexpr size_t kHeaderBytes = 4096;

struct Header {
    std::atomic<uint64_t> magic;  // written last by the creator: an attacher spins on it
    uint32_t nranks, narea;
    uint64_t block_doubles;
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> abort;
};
struct alignas(64) Line {
    std::atomic<uint64_t> ready;
    std::atomic<uint64_t> consumed;
};
static_assert(sizeof(Line) == 64, "one cache line per (area, rank)");
static_assert(std::atomic<uint64_t>::is_always_lock_free, "sequence words must be plain 8-byte atomics");

inline double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

class Gather {
public:
    int nranks = 0, rank = 0, narea = 0;
    size_t block_doubles = 0;
    double timeout_s = 60.0;

    static size_t bytes_for(int nranks, int narea, size_t block_doubles) {
        return kHeaderBytes + (size_t)narea * nranks * sizeof(Line) + (size_t)narea * nranks * block_doubles * sizeof(double);
    }
    // An extra region behind the blocks, on a page boundary of its own (the engine registers it with the HIP runtime: the
    // score board of the sharded device loop, which the GPUs of all ranks read and write -- muse_comm.cpp); zero-filled.
    static size_t extra_offset(int nranks, int narea, size_t block_doubles) {
        return (bytes_for(nranks, narea, block_doubles) + kHeaderBytes - 1) / kHeaderBytes * kHeaderBytes;
    }
    void* extra() const { return extra_bytes_ ? (void*)(base_ + extra_offset(nranks, narea, block_doubles)) : nullptr; }
    size_t extra_bytes() const { return extra_bytes_; }
    // Rank 0 creates the segment (O_EXCL), the others attach (retrying until it exists and its magic is set); the
    // creator unlinks the name once everyone has attached, so that nothing outlives the processes.
    bool open(const char* name, int nranks_, int rank_, int narea_, size_t block_doubles_, std::string& err, size_t extra_bytes = 0) {
        nranks = nranks_; rank = rank_; narea = narea_; block_doubles = block_doubles_;
        extra_bytes_ = (extra_bytes + kHeaderBytes - 1) / kHeaderBytes * kHeaderBytes;
        size_ = extra_bytes_ ? extra_offset(nranks, narea, block_doubles) + extra_bytes_ : bytes_for(nranks, narea, block_doubles);
        const double t0 = now_s();
        int fd = -1;
        if (rank == 0) {
            fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0) { err = std::string("shm_open(create ") + name + "): " + strerror(errno); return false; }
            if (ftruncate(fd, (off_t)size_) != 0) {
                err = std::string("ftruncate: ") + strerror(errno);
                ::close(fd); shm_unlink(name);
                return false;
            }
        } else {
            for (;;) {
                fd = shm_open(name, O_RDWR, 0600);
                if (fd >= 0) {
                    struct stat sb;
                    if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= size_) break;  // created AND sized
                    ::close(fd);
                    fd = -1;
                } else if (errno != ENOENT) {
                    err = std::string("shm_open(") + name + "): " + strerror(errno);
                    return false;
                }
                if (now_s() - t0 > timeout_s) { err = std::string("timed out waiting for segment ") + name; return false; }
                usleep(200);
            }
        }
        void* p = mmap(nullptr, size_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) {
            err = std::string("mmap: ") + strerror(errno);
            if (rank == 0) shm_unlink(name);
            return false;
        }
        base_ = (char*)p;
        hdr_ = (Header*)base_;
        lines_ = (Line*)(base_ + kHeaderBytes);
        data_ = (double*)(base_ + kHeaderBytes + (size_t)narea * nranks * sizeof(Line));
        if (rank == 0) {  // a fresh tmpfs segment is zero-filled: sequence words start at 0
            hdr_->nranks = (uint32_t)nranks;
            hdr_->narea = (uint32_t)narea;
            hdr_->block_doubles = block_doubles;
            hdr_->attached.store(1, std::memory_order_relaxed);
            hdr_->magic.store(kMagic, std::memory_order_release);
        } else {
            while (hdr_->magic.load(std::memory_order_acquire) != kMagic) {
                if (now_s() - t0 > timeout_s) { err = "timed out waiting for the creator of the segment"; close(); return false; }
                usleep(100);
            }
            if ((int)hdr_->nranks != nranks || (int)hdr_->narea != narea || hdr_->block_doubles != block_doubles) {
                err = "segment geometry differs from this rank's arguments";
                close();
                return false;
            }
            hdr_->attached.fetch_add(1, std::memory_order_acq_rel);
        }
        // everyone waits for everyone: after this the name is no longer needed
        while ((int)hdr_->attached.load(std::memory_order_acquire) < nranks) {
            if (now_s() - t0 > timeout_s || hdr_->abort.load(std::memory_order_relaxed)) {
                err = "timed out waiting for all ranks to attach";
                if (rank == 0) shm_unlink(name);
                close();
                return false;
            }
            usleep(100);
        }
        if (rank == 0) shm_unlink(name);
        return true;
    }
    void close() {
        if (base_ && !adopted_) munmap(base_, size_);
        base_ = nullptr; hdr_ = nullptr; lines_ = nullptr; data_ = nullptr;
        adopted_ = false;   // (a later open() maps a segment of this object's own; the owner of an adopted view must outlive it)
    }
    // A segment of this process alone (anonymous shared mapping): what the ranks-as-threads sanitizer test exchanges through.
    bool open_private(int nranks_, int narea_, size_t block_doubles_) {
        nranks = nranks_; rank = 0; narea = narea_; block_doubles = block_doubles_;
        size_ = bytes_for(nranks, narea, block_doubles);
        void* p = mmap(nullptr, size_, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return false;
        base_ = (char*)p;
        hdr_ = (Header*)base_;
        lines_ = (Line*)(base_ + kHeaderBytes);
        data_ = (double*)(base_ + kHeaderBytes + (size_t)narea * nranks * sizeof(Line));
        hdr_->nranks = (uint32_t)nranks;
        hdr_->narea = (uint32_t)narea;
        hdr_->block_doubles = block_doubles;
        hdr_->attached.store((uint32_t)nranks, std::memory_order_relaxed);
        hdr_->magic.store(kMagic, std::memory_order_release);
        return true;
    }
    // Another rank's view of a segment that `owner` has mapped, through the SAME mapping (the ranks of the sanitizer test are
    // threads of one process: ThreadSanitizer follows a word by its address, and two mappings of one page are two addresses).
    void adopt(const Gather& owner, int rank_) {
        close();
        nranks = owner.nranks; rank = rank_; narea = owner.narea; block_doubles = owner.block_doubles; timeout_s = owner.timeout_s;
        base_ = owner.base_; hdr_ = owner.hdr_; lines_ = owner.lines_; data_ = owner.data_; size_ = owner.size_;
        adopted_ = true;
    }
    ~Gather() { close(); }

    double* block(int area, int r) const { return data_ + ((size_t)area * nranks + r) * block_doubles; }
    void raise_abort() { if (hdr_) hdr_->abort.store(1, std::memory_order_release); }
    bool aborted() const { return hdr_ && hdr_->abort.load(std::memory_order_acquire) != 0; }
    int attached() const { return hdr_ ? (int)hdr_->attached.load(std::memory_order_acquire) : 0; }  // ranks that mapped the segment

    // 0 ok, 1 timeout, 2 aborted by a peer
    int wait_consumed(int area, uint64_t seq) const { return wait_all(area, seq, false); }
    int wait_ready(int area, uint64_t seq) const { return wait_all(area, seq, true); }
    void publish_ready(int area, uint64_t seq) { line(area, rank).ready.store(seq, std::memory_order_release); }
    void publish_consumed(int area, uint64_t seq) { line(area, rank).consumed.store(seq, std::memory_order_release); }

private:
    char* base_ = nullptr;
    size_t extra_bytes_ = 0;
    Header* hdr_ = nullptr;
    Line* lines_ = nullptr;
    double* data_ = nullptr;
    size_t size_ = 0;
    bool adopted_ = false;
    Line& line(int area, int r) const { return lines_[(size_t)area * nranks + r]; }
    int wait_all(int area, uint64_t seq, bool ready) const {
        double t0 = 0.0;
        for (int q = 0; q < nranks; ++q) {
            const std::atomic<uint64_t>& w = ready ? line(area, q).ready : line(area, q).consumed;
            unsigned spins = 0;
            while (w.load(std::memory_order_acquire) < seq) {
                MUSE_CPU_RELAX();
                if ((++spins & 0x3ff) == 0) {  // every ~1024 polls: abort word, clock, and let an oversubscribed host run the peer
                    if (hdr_->abort.load(std::memory_order_acquire)) return 2;
                    const double t = now_s();
                    if (t0 == 0.0) t0 = t;
                    if (t - t0 > timeout_s) return 1;
                    if (spins > (1u << 16)) sched_yield();
                }
            }
        }
        return 0;
    }
};

}  // namespace muse_shm
