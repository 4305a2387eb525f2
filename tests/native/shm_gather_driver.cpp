// Test driver for museinference.jl_amd/csrc/shm_gather.hpp (the shared-memory transport's protocol, no GPU):
//   shm_gather_driver <segment name> <nranks> <rank> <rounds> <block_doubles> [abort_at_round]
// Every rank runs `rounds` exchanges over 4 areas in turn with rank- and round-dependent delays, fills its block with
// a function of (rank, area, sequence number, index) and checks every block it reads.  With abort_at_round the LAST
// rank raises the abort word instead of publishing; the others must then leave their wait with code 2 (exit 42).
//   shm_gather_driver --threads <segment name> <nranks> <rounds> <block_doubles>
// The same exchange with the ranks as THREADS of this process sharing one mapping of the segment -- the form
// ThreadSanitizer can follow (tests/test_sanitizers.py).
#include <stdio.h>
#include <stdlib.h>

#include <thread>
#include <vector>

#include "../../museinference.jl_amd/csrc/shm_gather.hpp"

static double value(int rank, int area, uint64_t seq, size_t i) {
    return (double)rank * 1e6 + (double)area * 1e5 + (double)(seq % 100000) + 1e-3 * (double)(i % 997);
}

static int exchange(muse_shm::Gather& g, int nranks, int rank, int rounds, size_t B, int abort_at);

int main(int argc, char** argv) {
    if (argc >= 6 && std::string(argv[1]) == "--threads") {
        const int nranks = atoi(argv[3]), rounds = atoi(argv[4]);   // (argv[2], the segment name, is not used: the mapping is private)
        const size_t B = (size_t)atol(argv[5]);
        muse_shm::Gather owner;
        owner.timeout_s = 60.0;
        if (!owner.open_private(nranks, 4, B)) {
            fprintf(stderr, "open_private failed\n");
            return 3;
        }
        std::vector<muse_shm::Gather> views((size_t)nranks);
        std::vector<int> rc((size_t)nranks, -1);
        std::vector<std::thread> th;
        for (int r = 0; r < nranks; ++r) views[(size_t)r].adopt(owner, r);
        for (int r = 0; r < nranks; ++r)
            th.emplace_back([&, r] { rc[(size_t)r] = exchange(views[(size_t)r], nranks, r, rounds, B, -1); });
        for (auto& t : th) t.join();
        for (int r = 0; r < nranks; ++r)
            if (rc[(size_t)r]) return rc[(size_t)r];
        return 0;
    }
    if (argc < 6) return 2;
    const char* name = argv[1];
    const int nranks = atoi(argv[2]), rank = atoi(argv[3]), rounds = atoi(argv[4]);
    const size_t B = (size_t)atol(argv[5]);
    const int abort_at = argc > 6 ? atoi(argv[6]) : -1;
    muse_shm::Gather g;
    g.timeout_s = 20.0;
    std::string err;
    // MUSE_TEST_EXTRA=bytes: the segment carries an extra region behind the blocks (the score board of the sharded device loop,
    // muse_comm.cpp): page-aligned, zero-filled, the same memory in every rank -- a word every rank writes before the exchanges is
    // what every rank reads after them
    const size_t extra = getenv("MUSE_TEST_EXTRA") ? (size_t)atol(getenv("MUSE_TEST_EXTRA")) : 0;
    if (!g.open(name, nranks, rank, 4, B, err, extra)) {
        fprintf(stderr, "rank %d: open failed: %s\n", rank, err.c_str());
        return 3;
    }
    volatile uint64_t* board = (volatile uint64_t*)g.extra();
    if (extra) {
        if (!board || ((uintptr_t)board & 4095) != 0 || g.extra_bytes() < extra || g.extra_bytes() % 4096 != 0) return 5;
        if (board[(size_t)nranks * 8 + 1] != 0) return 6;   // (a fresh segment is zero-filled)
        board[(size_t)rank * 8] = 0xABCD0000ull + (uint64_t)rank;
    } else if (board) {
        return 5;
    }
    const int rc = exchange(g, nranks, rank, rounds, B, abort_at);
    if (rc == 0 && extra)
        for (int q = 0; q < nranks; ++q)
            if (board[(size_t)q * 8] != 0xABCD0000ull + (uint64_t)q) return 7;
    return rc;
}

static int exchange(muse_shm::Gather& g, int nranks, int rank, int rounds, size_t B, int abort_at) {
    uint64_t seq[4] = {0, 0, 0, 0};
    unsigned lcg = 12345u + 77u * (unsigned)rank;
    for (int k = 0; k < rounds; ++k) {
        const int area = k % 4;
        const uint64_t s = ++seq[area];
        lcg = lcg * 1664525u + 1013904223u;
        if ((lcg >> 24) % 8 == (unsigned)rank % 8) usleep((lcg >> 16) % 300);  // a different straggler every few rounds
        int w = g.wait_consumed(area, s - 1);
        if (w) { fprintf(stderr, "rank %d round %d: wait_consumed -> %d\n", rank, k, w); return w == 2 ? 42 : 4; }
        if (k == abort_at && rank == nranks - 1) {
            g.raise_abort();
            return 43;
        }
        double* mine = g.block(area, rank);
        for (size_t i = 0; i < B; ++i) mine[i] = value(rank, area, s, i);
        g.publish_ready(area, s);
        w = g.wait_ready(area, s);
        if (w) { fprintf(stderr, "rank %d round %d: wait_ready -> %d\n", rank, k, w); return w == 2 ? 42 : 5; }
        for (int q = 0; q < nranks; ++q) {
            const double* b = g.block(area, q);
            for (size_t i = 0; i < B; ++i)
                if (b[i] != value(q, area, s, i)) {
                    fprintf(stderr, "rank %d round %d: block of rank %d, word %zu: %.17g != %.17g\n", rank, k, q, i, b[i],
                            value(q, area, s, i));
                    return 6;
                }
        }
        g.publish_consumed(area, s);
    }
    printf("rank %d ok %d rounds\n", rank, rounds);
    return 0;
}
