// pcr_hash64: a fast 64-bit content hash of a host buffer (host code only).
//
// Why it is in this library: the reference's calc_H_g_e2(cur_T, source) takes the scan as a NumPy array on every
// call (registration.py:55-68) and is pure in it.  The drop-in class keeps the device copy of the last scan and
// must notice ANY edit of the caller's array, so it hashes the whole buffer per call -- 1.4 ms with xxh3 in
// Python for 1.06 M points (11 ms with the crc32 fallback), 9-70x the 0.16 ms GPU pass behind it.
// Here: fixed 256 KiB chunks hashed in parallel by a small persistent pool of threads (wyhash-style 64x64->128
// multiply-mix, four independent lanes per chunk), chunk digests folded in order -- the value does not depend on
// the number of threads.  Not cryptographic; 64 bits: a collision needs ~2^32 distinct scans.
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "pcr.h"

namespace {

inline uint64_t mum(uint64_t a, uint64_t b) {
    const __uint128_t r = (__uint128_t)a * b;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
}
inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

const uint64_t K0 = 0xa0761d6478bd642full, K1 = 0xe7037ed1a0b428dbull, K2 = 0x8ebc6af09c88c6e3ull, K3 = 0x589965cc75374cc3ull;

uint64_t hash_chunk(const uint8_t *p, size_t n, uint64_t seed) {
    uint64_t s0 = seed ^ K0, s1 = seed ^ K1, s2 = seed ^ K2, s3 = seed ^ K3;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        s0 = mum(rd64(p + i) ^ K1, rd64(p + i + 8) ^ s0);
        s1 = mum(rd64(p + i + 16) ^ K2, rd64(p + i + 24) ^ s1);
        s2 = mum(rd64(p + i + 32) ^ K3, rd64(p + i + 40) ^ s2);
        s3 = mum(rd64(p + i + 48) ^ K0, rd64(p + i + 56) ^ s3);
    }
    uint8_t tail[64];
    const size_t r = n - i;
    if (r) {
        memset(tail, 0, sizeof tail);
        memcpy(tail, p + i, r);
        s0 = mum(rd64(tail) ^ K1, rd64(tail + 8) ^ s0);
        s1 = mum(rd64(tail + 16) ^ K2, rd64(tail + 24) ^ s1);
        s2 = mum(rd64(tail + 32) ^ K3, rd64(tail + 40) ^ s2);
        s3 = mum(rd64(tail + 48) ^ K0, rd64(tail + 56) ^ s3);
    }
    return mum(s0 ^ s2 ^ (uint64_t)n, s1 ^ s3 ^ K2);
}

const size_t CHUNK = (size_t)256 << 10;

struct Pool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    uint64_t generation = 0;
    int busy = 0;
    bool quit = false;
    // the job
    const uint8_t *data = nullptr;
    size_t bytes = 0, nchunks = 0;
    std::atomic<size_t> next{0};
    std::vector<uint64_t> digests;

    void work() {
        for (;;) {
            const size_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= nchunks) break;
            const size_t off = c * CHUNK;
            digests[c] = hash_chunk(data + off, bytes - off < CHUNK ? bytes - off : CHUNK, (uint64_t)c);
        }
    }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_go.wait(lk, [&] { return quit || generation != seen; });
            if (quit) return;
            seen = generation;
            lk.unlock();
            work();
            lk.lock();
            if (--busy == 0) cv_done.notify_one();
        }
    }
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) workers.emplace_back([this] { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_go.notify_all();
        for (auto &t : workers) t.join();
    }
    uint64_t run(const uint8_t *p, size_t n) {
        data = p; bytes = n; nchunks = (n + CHUNK - 1) / CHUNK;
        digests.assign(nchunks, 0);
        next.store(0);
        const bool par = nchunks >= 4 && !workers.empty();
        if (par) {
            { std::lock_guard<std::mutex> lk(m); busy = (int)workers.size(); ++generation; }
            cv_go.notify_all();
        }
        work();                                     // the caller hashes chunks too
        if (par) {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return busy == 0; });
        }
        uint64_t h = mum((uint64_t)n ^ K3, K0);
        for (size_t c = 0; c < nchunks; ++c) h = mum(h ^ digests[c], K1 ^ (uint64_t)c);
        return h;
    }
};

std::mutex g_pool_mutex;
Pool *g_pool = nullptr;

// fork(): the child inherits g_pool but none of its threads (the next hash would wait for workers that do not exist),
// and possibly g_pool_mutex locked by a thread that does not exist either.  The child starts over: a fresh
// mutex, no pool (the parent's Pool object is leaked in the child -- its threads cannot be joined there).
void hash_atfork_child() {
    new (&g_pool_mutex) std::mutex();
    g_pool = nullptr;
}
std::once_flag g_atfork_once;

// CPUs this process may use: the affinity mask capped by the cgroup CPU-bandwidth quota (cpu.max of cgroup v2,
// cfs_quota_us of v1).  A GPU box shows 256 CPUs to a container that may run 16: a pool sized by
// hardware_concurrency() burns the quota and gets every thread of the process parked.
int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = c; }
    double quota = -1.0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; double per = 0;
        if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / per;
        fclose(f);
    } else if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        double qv = -1, per = 0;
        if (fscanf(fq, "%lf", &qv) != 1) qv = -1;
        fclose(fq);
        if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(fp, "%lf", &per) != 1) per = 0;
            fclose(fp);
        }
        if (qv > 0 && per > 0) quota = qv / per;
    }
    if (quota > 0 && quota < n) n = (int)quota;
    return n < 1 ? 1 : n;
}

}  // namespace

// (exported for tests and for INTEGRATION.md's advice on sizing host thread pools)
extern "C" int pcr_usable_cpus(void) { return usable_cpus(); }

extern "C" pcr_status pcr_hash64(const void *data, uint64_t nbytes, uint64_t *out) {
    if (!out || (!data && nbytes)) return PCR_ERR_INVALID;
    std::call_once(g_atfork_once, [] { (void)pthread_atfork(nullptr, nullptr, hash_atfork_child); });
    std::lock_guard<std::mutex> lk(g_pool_mutex);           // one hash at a time per process
    if (!g_pool) {
        const int hw = usable_cpus();                       // the CPU quota, not the visible CPUs
        int n = hw >= 32 ? 15 : (hw >= 16 ? 11 : (hw >= 8 ? 7 : (hw >= 2 ? hw - 1 : 0)));
        const char *e = getenv("PCR_HASH_THREADS");           // (developer: worker threads beside the caller)
        if (e && *e) { n = atoi(e); if (n < 0) n = 0; if (n > 63) n = 63; }
        g_pool = new Pool(n);                               // lives until the process exits
    }
    *out = g_pool->run((const uint8_t *)data, (size_t)nbytes);
    return PCR_OK;
}

// ---- LZF (the codec of PCD "DATA binary_compressed"; round 6: io.py's byte loop in Python took minutes per 1e6 points) --------
// Format (liblzf): control byte c < 32: c + 1 literal bytes follow; else a back reference of length (c >> 5) + 2 (c >> 5 == 7: + the
// next byte) at distance ((c & 31) << 8 | next byte) + 1 behind the write position; copies may overlap their own output.
extern "C" pcr_status pcr_lzf_decompress(const void *in_, uint64_t in_len, void *out_, uint64_t out_len, uint64_t *written) {
    if ((!in_ && in_len) || (!out_ && out_len) || !written) return PCR_ERR_INVALID;
    const uint8_t *in = (const uint8_t *)in_;
    uint8_t *out = (uint8_t *)out_;
    uint64_t ip = 0, op = 0;
    *written = 0;
    while (ip < in_len) {
        const unsigned c = in[ip++];
        if (c < 32) {
            const uint64_t ln = c + 1;
            if (ip + ln > in_len || op + ln > out_len) return PCR_ERR_INVALID;
            memcpy(out + op, in + ip, ln);
            ip += ln; op += ln;
        } else {
            uint64_t ln = c >> 5;
            if (ln == 7) { if (ip >= in_len) return PCR_ERR_INVALID; ln += in[ip++]; }
            if (ip >= in_len) return PCR_ERR_INVALID;
            const uint64_t dist = (((uint64_t)(c & 0x1f)) << 8 | in[ip++]) + 1;
            ln += 2;
            if (dist > op || op + ln > out_len) return PCR_ERR_INVALID;
            const uint8_t *src = out + op - dist;
            if (dist >= ln) memcpy(out + op, src, ln);
            else for (uint64_t i = 0; i < ln; ++i) out[op + i] = src[i];      // overlapping: byte by byte, forwards
            op += ln;
        }
    }
    *written = op;
    return PCR_OK;
}

// greedy single-pass compressor (save_pcd(..., compressed=True) and the tests' 1e6-point files); out_cap >= in_len + in_len / 32 + 8
// always suffices (all literals).  Not liblzf's bit stream, but a valid one: every LZF decoder reads it.
extern "C" pcr_status pcr_lzf_compress(const void *in_, uint64_t in_len, void *out_, uint64_t out_cap, uint64_t *written) {
    if ((!in_ && in_len) || !out_ || !written) return PCR_ERR_INVALID;
    const uint8_t *in = (const uint8_t *)in_;
    uint8_t *out = (uint8_t *)out_;
    std::vector<uint64_t> table((size_t)1 << 16, ~(uint64_t)0);
    uint64_t ip = 0, op = 0, lit = 0;             // lit: literals pending since position ip - lit
    auto flush = [&](uint64_t upto) -> bool {
        uint64_t s = upto - lit;
        while (lit) {
            const uint64_t n = lit < 32 ? lit : 32;
            if (op + 1 + n > out_cap) return false;
            out[op++] = (uint8_t)(n - 1);
            memcpy(out + op, in + s, n);
            op += n; s += n; lit -= n;
        }
        return true;
    };
    while (ip < in_len) {
        uint64_t mlen = 0, ref = 0;
        if (ip + 3 <= in_len) {
            const uint32_t h = ((uint32_t)in[ip] << 16 | (uint32_t)in[ip + 1] << 8 | in[ip + 2]) * 2654435761u >> 16;
            ref = table[h];
            table[h] = ip;
            if (ref != ~(uint64_t)0 && ip - ref <= 8192 && in[ref] == in[ip] && in[ref + 1] == in[ip + 1] && in[ref + 2] == in[ip + 2]) {
                const uint64_t maxl = in_len - ip < 264 ? in_len - ip : 264;
                mlen = 3;
                while (mlen < maxl && in[ref + mlen] == in[ip + mlen]) ++mlen;
            }
        }
        if (mlen >= 3) {
            if (!flush(ip)) return PCR_ERR_INVALID;
            const uint64_t dist = ip - ref - 1, l = mlen - 2;
            if (op + 3 > out_cap) return PCR_ERR_INVALID;
            if (l < 7) out[op++] = (uint8_t)((l << 5) | (dist >> 8));
            else { out[op++] = (uint8_t)((7u << 5) | (dist >> 8)); out[op++] = (uint8_t)(l - 7); }
            out[op++] = (uint8_t)(dist & 0xff);
            ip += mlen;
        } else {
            ++lit; ++ip;
        }
    }
    if (!flush(ip)) return PCR_ERR_INVALID;
    *written = op;
    return PCR_OK;
}
