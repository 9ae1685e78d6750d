// geosrad.hip -- C-ABI (include/geosrad.h) of the MI355X-native radiation hot path: context, coefficient
// table upload (GRTB blobs -> GPU-friendly layouts), HBM workspace, kernel launches.
// There is deliberately NO CPU fallback: without a HIP device geosrad_create() fails with GEOSRAD_ENODEV.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <memory>
#include <atomic>
#include <functional>
#include <chrono>

#include <dlfcn.h>
#include <sched.h>
#include <cstddef>
#include <cstdint>
#include "../../include/geosrad.h"
#include "lw_device.hpp"
#include "lw_kernels.hpp"
#include "sw_kernels.hpp"
#include "mcica_kernels.hpp"
#include "chou_kernels.hpp"
#include "sorad_kernels.hpp"
#include "gridcomp_kernels.hpp"
#include "lw_cols.hpp"
#include "lw_split.hpp"
#include "sw_reform.hpp"

using namespace geosrad;

namespace {

struct BlobEntry { int kind, ndim, dims[4]; const char *data; size_t count; };
struct Blob {
    int realbytes = 0;
    std::map<std::string, BlobEntry> e;
    std::string err;
    bool parse(const void *blob, size_t nbytes)
    {
        const char *b = (const char *)blob;
        if (nbytes < 12 || memcmp(b, "GRTB", 4) != 0) { err = "not a GRTB blob"; return false; }
        int32_t ver, rb;
        memcpy(&ver, b + 4, 4); memcpy(&rb, b + 8, 4);
        if (ver != 1 || (rb != 4 && rb != 8)) { err = "unsupported GRTB version / real size"; return false; }
        realbytes = rb;
        size_t off = 12;
        while (true) {
            if (off + 56 > nbytes) { err = "truncated GRTB blob"; return false; }
            char name[33]; memcpy(name, b + off, 32); name[32] = 0;
            int32_t h[6]; memcpy(h, b + off + 32, 24);
            off += 56;
            if (!strcmp(name, "END")) break;
            BlobEntry en; en.kind = h[0]; en.ndim = h[1];
            size_t cnt = 1;
            for (int k = 0; k < 4; k++) { en.dims[k] = h[2 + k]; if (k < en.ndim) cnt *= (size_t)h[2 + k]; }
            en.count = cnt; en.data = b + off;
            size_t nb = cnt * (size_t)(en.kind < 0 ? -en.kind : en.kind);
            if (off + nb > nbytes) { err = "truncated GRTB blob"; return false; }
            off += nb + ((8 - nb % 8) % 8);
            e[name] = en;
        }
        return true;
    }
};

#define HIPCHK(call)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) { return fail(GEOSRAD_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } \
    } while (0)

// ---- KISS jump-ahead constants (see mcica_kernels.hpp: KissJump) --------------------------------------------
static uint32_t xs_step(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
static uint32_t mat_apply(const uint32_t *cols, uint32_t v)
{
    uint32_t y = 0;
    for (int i = 0; i < 32; i++) if ((v >> i) & 1u) y ^= cols[i];
    return y;
}
static uint32_t powmod(uint64_t a, uint64_t n, uint64_t m)
{
    uint64_t r = 1 % m; a %= m;
    while (n) { if (n & 1) r = r * a % m; a = a * a % m; n >>= 1; }
    return (uint32_t)r;
}
static KissJump make_kiss_jump(uint64_t n)
{
    KissJump J;
    // LCG: compose (A,C) by binary exponentiation of x -> a x + c
    uint32_t A = 1, C = 0, a = 69069u, c = 1327217885u;
    for (uint64_t k = n; k; k >>= 1) {
        if (k & 1) { A = A * a; C = C * a + c; }
        c = c * a + c; a = a * a;          // (a,c) o (a,c)
    }
    J.A1 = A; J.C1 = C;
    // xorshift matrix power
    uint32_t R[32], P[32], t[32];
    for (int i = 0; i < 32; i++) { R[i] = 1u << i; P[i] = xs_step(1u << i); }
    for (uint64_t k = n; k; k >>= 1) {
        if (k & 1) { for (int i = 0; i < 32; i++) t[i] = mat_apply(P, R[i]); memcpy(R, t, sizeof t); }
        for (int i = 0; i < 32; i++) t[i] = mat_apply(P, P[i]);
        memcpy(P, t, sizeof t);
    }
    memcpy(J.M2, R, sizeof R);
    J.K3 = powmod(18000u, n, 18000ull * 65536ull - 1ull);
    J.K4 = powmod(30903u, n, 30903ull * 65536ull - 1ull);
    return J;
}
// staging of one coefficient blob into GPU-friendly layouts: k-table rows [index][NGP = pad4(ng)] so that a lane
// fetches 4 consecutive g-points of its own row with one 16-byte load
template <typename R> struct TableStage {
    const Blob &B;
    std::vector<char> stage;
    std::vector<std::pair<const R **, size_t>> fix;   // (pointer slot, byte offset)
    std::string missing;
    explicit TableStage(const Blob &b) : B(b) {}
    const R *get(const std::string &nm, size_t expect)
    {
        auto it = B.e.find(nm);
        if (it == B.e.end() || it->second.kind != (int)sizeof(R) || (expect && it->second.count != expect)) {
            missing += nm + " ";
            return nullptr;
        }
        return (const R *)it->second.data;
    }
    size_t reserve(size_t nreal)
    {
        size_t off = (stage.size() + 15) & ~(size_t)15;
        stage.resize(off + nreal * sizeof(R), 0);
        return off;
    }
    R *at(size_t off) { return (R *)(stage.data() + off); }
    // Fortran (n1, ng) -> [n1][NGP]
    void tr2(const R **slot, const std::string &nm, int n1, int ng)
    {
        const R *s = get(nm, (size_t)n1 * ng);
        if (!s) return;
        const int ngp = pad4(ng);
        size_t off = reserve((size_t)n1 * ngp);
        for (int g = 0; g < ng; g++)
            for (int i = 0; i < n1; i++) at(off)[(size_t)i * ngp + g] = s[(size_t)g * n1 + i];
        fix.push_back({slot, off});
    }
    // Fortran (nsp, 19, ng) -> [19][nsp][NGP]
    void tr3(const R **slot, const std::string &nm, int nsp, int ng)
    {
        const R *s = get(nm, (size_t)nsp * 19 * ng);
        if (!s) return;
        const int ngp = pad4(ng);
        size_t off = reserve((size_t)19 * nsp * ngp);
        for (int g = 0; g < ng; g++)
            for (int im = 0; im < 19; im++)
                for (int j = 0; j < nsp; j++) at(off)[((size_t)im * nsp + j) * ngp + g] = s[((size_t)g * 19 + im) * nsp + j];
        fix.push_back({slot, off});
    }
    // Fortran (ng, m) -> [m][NGP]   (m = 1 for plain per-g vectors)
    void rows(const R **slot, const std::string &nm, int ng, int m)
    {
        const R *s = get(nm, (size_t)ng * m);
        if (!s) return;
        const int ngp = pad4(ng);
        size_t off = reserve((size_t)m * ngp);
        for (int j = 0; j < m; j++)
            for (int g = 0; g < ng; g++) at(off)[(size_t)j * ngp + g] = s[(size_t)j * ng + g];
        fix.push_back({slot, off});
    }
    // one scalar replicated over a [NGP] row
    void splat(const R **slot, const std::string &nm, int ng)
    {
        const R *s = get(nm, 1);
        if (!s) return;
        const int ngp = pad4(ng);
        size_t off = reserve((size_t)ngp);
        for (int g = 0; g < ngp; g++) at(off)[g] = *s;
        fix.push_back({slot, off});
    }
    void raw(const R **slot, const std::string &nm, size_t cnt)
    {
        const R *s = get(nm, cnt);
        if (!s) return;
        size_t off = reserve(cnt);
        memcpy(at(off), s, cnt * sizeof(R));
        fix.push_back({slot, off});
    }
    R scalar(const std::string &nm) { const R *s = get(nm, 1); return s ? *s : (R)0; }
    bool ints(const std::string &nm, size_t cnt, int32_t *dst)
    {
        auto it = B.e.find(nm);
        if (it == B.e.end() || it->second.kind != -4 || it->second.count != cnt) { missing += nm + " "; return false; }
        memcpy(dst, it->second.data, cnt * sizeof(int32_t));
        return true;
    }
};

// roctx ranges with the reference's MAPL timer names around the kernel groups (the names of GEOS_IrradGridComp.F90:1138-1155 and
// rrtmg_sw_rad.F90:1181-1200 / rrtmg_sw_spcvmc.F90:382-567 survive on the GPU timeline: `rocprofv3 --marker-trace`).  Off unless
// GEOSRAD_ROCTX=1; the marker library is looked up at run time (no link dependency).  A fused kernel group carries the names of all the
// reference stages it covers, nested.
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi()
    {
        const char *e = getenv("GEOSRAD_ROCTX");
        if (!e || atoi(e) == 0) return;
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
static RoctxApi &roctx_api() { static RoctxApi a; return a; }
// kernel-group id (geosrad_kernel_name) -> the reference timers it stands for (up to three, outermost first)
static const char *const ROCTX_NAMES[14][3] = {
    {"---RRTMG_RUN", "k_validate_pwv", nullptr}, {"---RRTMG_RUN", "setcoef", nullptr}, {"---RRTMG_CLDSGEN", "overlap", nullptr},
    {"---RRTMG_CLDSGEN", "---RRTMG_CLDPRMC", nullptr}, {"---RRTMG_RUN", "taumol+rtrnmc", nullptr}, {"---RRTMG_RUN", "reduce", nullptr},
    {"---RRTMG_PART", nullptr, nullptr}, {"---RRTMG_SETCOEF", nullptr, nullptr}, {"---RRTMG_TAUMOL", "---RRTMG_REFTRA", "---RRTMG_VRTQDR"},
    {"---RRTMG_PART", "reduce", nullptr}, {"---IRRAD_RUN", "prep", nullptr}, {"---IRRAD_RUN", "bands", nullptr}, {"---SORAD_RUN", "prep", nullptr},
    {"---SORAD_RUN", "passes", nullptr}};

static const char *LW_NEG_NAMES[21] = {"play", "tlay", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr",
                                       "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr", "cldf", "ciwp", "clwp", "rei", "rel",
                                       "plev", "tsfc", "emis", "tauaer"};

}  // namespace

// The copy threads of the host-pointer entry points: created once per context (first chunk that is worth splitting) and parked on a
// condition variable between chunks - a chunk's gather / scatter is a few milliseconds, a std::thread spawn + join per chunk and
// direction was a measurable part of it
class CopyPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::function<void(size_t, size_t)> *job = nullptr;
    size_t nitems = 0, per = 0;
    int pending = 0;
    unsigned long gen = 0;
    bool quit = false;
    void loop(int t)
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(size_t, size_t)> *f; size_t lo, hi;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen; f = job;
                lo = (size_t)t * per; hi = lo + per < nitems ? lo + per : nitems;
            }
            if (lo < hi) (*f)(lo, hi);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
public:
    explicit CopyPool(int n) { for (int t = 0; t < n; t++) th.emplace_back([this, t] { loop(t); }); }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
    int size() const { return (int)th.size(); }
    // fn(lo, hi) over [0, n) cut into size() contiguous pieces; returns when all pieces are done
    void run(size_t n, const std::function<void(size_t, size_t)> &fn)
    {
        std::unique_lock<std::mutex> lk(mu);
        job = &fn; nitems = n; per = (n + th.size() - 1) / th.size(); pending = (int)th.size(); gen++;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

// copy threads per context when GEOSRAD_HOST_THREADS is not set: the host's hardware threads shared out among the ranks of the node (the
// launchers' node-local size; 96 ranks x 8 copy threads would oversubscribe a host), at most 8, at least 1
// CPUs this process may use at once: the hardware threads it is allowed on, capped by the cgroup's CPU quota (a container that sees 256
// hardware threads may own 16 of them)
static int usable_cpus()
{
    long n = (long)std::thread::hardware_concurrency();
    cpu_set_t m;
    if (sched_getaffinity(0, sizeof m, &m) == 0 && CPU_COUNT(&m) > 0) n = CPU_COUNT(&m);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2: "<quota> <period>" or "max <period>"
        long q = 0, per = 0;
        if (fscanf(f, "%ld %ld", &q, &per) == 2 && q > 0 && per > 0) { const long c = (q + per - 1) / per; if (c < n) n = c; }
        fclose(f);
    }
    return (int)(n < 1 ? 1 : n);
}

static int default_host_threads()
{
    unsigned hw = (unsigned)usable_cpus();
    if (hw == 0) hw = 8;
    long ranks = 1;
    static const char *vars[] = {"OMPI_COMM_WORLD_LOCAL_SIZE", "MV2_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "PMI_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE"};
    for (const char *v : vars) {
        const char *e = getenv(v);
        if (!e || !*e) continue;
        char *end = nullptr;
        const long r = strtol(e, &end, 10);
        if (end != e && r >= 1) { ranks = r; break; }
    }
    long n = (long)hw / ranks;
    return (int)(n < 1 ? 1 : (n > 8 ? 8 : n));
}

// ---------------------------------------------------------------------------------------------------
struct geosrad_ctx {
    int device = 0, real_kind = 4, chunk = 131072;
    bool sorad_col_path = false;    // Chou-Suarez sorad passes: HBM scratch planes, lane = column (default) | GEOSRAD_SORAD_PATH=col: on chip
    bool lw_cols_path = false;      // RRTMG_LW band sweeps: parked cells in HBM (default) | GEOSRAD_LW_PATH=cols: on-chip intermediates
    bool lw_split_path = false;     //                       | GEOSRAD_LW_PATH=split: k-distribution layer-parallel (k_lw_cells) + recurrences (k_lw_sweep)
    int sw_path = 2;                // RRTMG_SW band sweeps: k_sw_reform (2, default) | GEOSRAD_SW_PATH=bands: k_sw_bands, the first mapping (0)
    std::string last_error;
    hipStream_t stream = nullptr;   // internal stream of the host-pointer entry points
    // optional per-kernel timing with HIP events recorded on the launch stream (geosrad_profile*)
    bool profiling = false;
    struct Span { int kid; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> evpool;
    double prof_ms[16] = {0};
    long prof_n[16] = {0};
    hipEvent_t getev()
    {
        hipEvent_t e = nullptr;
        if (!evpool.empty()) { e = evpool.back(); evpool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    }
    int roctx_depth = 0;
    void span_begin(int kid, hipStream_t st)
    {
        if (roctx_api().push && kid >= 0 && kid < 14) {
            for (int k = 0; k < 3; k++) if (ROCTX_NAMES[kid][k]) { roctx_api().push(ROCTX_NAMES[kid][k]); roctx_depth++; }
        }
        if (!profiling) return;
        Span s{kid, getev(), getev()};
        (void)hipEventRecord(s.a, st);
        spans.push_back(s);
    }
    void span_end(hipStream_t st)
    {
        if (profiling && !spans.empty()) (void)hipEventRecord(spans.back().b, st);
        for (; roctx_depth > 0; roctx_depth--) roctx_api().pop();
    }
    void prof_collect()
    {
        for (auto &s : spans) {
            float ms = 0;
            if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { prof_ms[s.kid] += ms; prof_n[s.kid]++; }
            evpool.push_back(s.a); evpool.push_back(s.b);
        }
        spans.clear();
    }
    int fail(int code, const std::string &msg) { last_error = msg; return code; }

    // ---- host-pointer entry points: chunk pipeline --------------------------------------------------------------------------
    // The reference interface hands over host arrays, Fortran (ncol, rows): column index fastest.  A batch goes through the GPU in
    // chunks of `host_chunk` columns: copy threads gather a chunk's rows into a pinned staging slot, one DMA moves the slot to HBM,
    // the solver runs on it, one DMA brings the outputs back and the threads scatter them - on three streams and two slots, so that
    // the gathering of chunk k+1, the transfers and the kernels of chunk k and the scattering of chunk k-1 overlap.
    struct PipeArr { const void *src; void *dst; size_t rows, ebytes; size_t off; };      // src: copied in; dst: copied back (either may be null)
    int host_chunk = 16384, host_chunk_default = 16384, host_threads = default_host_threads();
    std::unique_ptr<CopyPool> copy_pool;
    size_t host_ld = 0;             // leading dimension (columns) of the caller's arrays when the call covers a shard of them (multi-device context); 0: ncol
    bool host_nt = true;            // non-temporal stores into the staging slots (GEOSRAD_HOST_NT=0: plain memcpy)
    // three staging slots, results copied back to the caller two chunks behind the one being gathered: the host thread then never waits
    // for the GPU in steady state and the H2D engine always has the next chunk queued (two slots in lock-step left it idle while the
    // host gathered: 4.25 instead of 3.4 ms per 16 384-column chunk)
    static constexpr int PIPE_SLOTS = 3, PIPE_LAG = 2;
    static constexpr int PIPE_FLAGGED = -77;      // host_pipeline: a chunk came back with the device error word set (the caller's check() names it)
    char *pipe_pin[PIPE_SLOTS][2] = {};      // [slot][0 = to the device, 1 = from the device]
    char *pipe_dev[PIPE_SLOTS] = {};
    size_t pipe_pin_bytes[2] = {0, 0}, pipe_dev_bytes = 0;      // [0] holds a chunk's inputs, [1] its outputs only
    uint32_t *pipe_err = nullptr;            // pinned: the solver's device error word as of each slot's copy-back
    hipStream_t pipe_h2d = nullptr, pipe_d2h = nullptr;
    hipEvent_t pipe_ev[PIPE_SLOTS][3] = {};      // per slot: h2d, compute, d2h done
    void pipe_release()
    {
        for (int s = 0; s < PIPE_SLOTS; s++) {
            for (int d = 0; d < 2; d++) if (pipe_pin[s][d]) { (void)hipHostFree(pipe_pin[s][d]); pipe_pin[s][d] = nullptr; }
            if (pipe_dev[s]) { (void)hipFree(pipe_dev[s]); pipe_dev[s] = nullptr; }
            for (int e = 0; e < 3; e++) if (pipe_ev[s][e]) { (void)hipEventDestroy(pipe_ev[s][e]); pipe_ev[s][e] = nullptr; }
        }
        if (pipe_h2d) { (void)hipStreamDestroy(pipe_h2d); pipe_h2d = nullptr; }
        if (pipe_d2h) { (void)hipStreamDestroy(pipe_d2h); pipe_d2h = nullptr; }
        if (pipe_err) { (void)hipHostFree(pipe_err); pipe_err = nullptr; }
        pipe_pin_bytes[0] = pipe_pin_bytes[1] = pipe_dev_bytes = 0;
    }
    // a row into the write-once staging memory with non-temporal stores: no read-for-ownership of the destination lines (the rows,
    // 64 KB each, are below the size from which memcpy streams by itself)
    static void copy_stream(char *dst, const char *src, size_t n)
    {
        typedef long long v2a __attribute__((vector_size(16), aligned(16)));
        typedef long long v2u __attribute__((vector_size(16), aligned(1)));
        size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
        if (head > n) head = n;
        memcpy(dst, src, head); dst += head; src += head; n -= head;
        const size_t nv = n / 16;
        for (size_t i = 0; i < nv; i++) __builtin_nontemporal_store(*(const v2u *)(src + 16 * i), (v2a *)(dst + 16 * i));
        memcpy(dst + 16 * nv, src + 16 * nv, n - 16 * nv);
    }
    // rows of `arrs` (a chunk's nc columns starting at c0 of ncol) between the caller's arrays and a staging slot, on copy threads
    void pipe_copy(std::vector<PipeArr> &arrs, char *slot, size_t slot_base, int ncol, int c0, int nc, bool to_slot)
    {
        struct Item { const PipeArr *a; size_t row; };
        std::vector<Item> items;
        for (auto &a : arrs) {
            if (to_slot ? !a.src : !a.dst) continue;
            for (size_t r = 0; r < a.rows; r++) items.push_back({&a, r});
        }
        auto work = [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                const PipeArr &a = *items[i].a;
                const size_t r = items[i].row;
                char *sl = slot + (a.off - slot_base) + r * (size_t)nc * a.ebytes;      // a chunk's arrays are dense: leading dimension nc
                if (to_slot) {
                    if (host_nt) copy_stream(sl, (const char *)a.src + (r * (size_t)ncol + (size_t)c0) * a.ebytes, (size_t)nc * a.ebytes);
                    else memcpy(sl, (const char *)a.src + (r * (size_t)ncol + (size_t)c0) * a.ebytes, (size_t)nc * a.ebytes);
                } else memcpy((char *)a.dst + (r * (size_t)ncol + (size_t)c0) * a.ebytes, sl, (size_t)nc * a.ebytes);
            }
            if (to_slot && host_nt) std::atomic_thread_fence(std::memory_order_seq_cst);      // the streamed rows are visible before the DMA reads them
        };
        size_t bytes = 0;
        for (auto &it : items) bytes += (size_t)nc * it.a->ebytes;
        const int nt = bytes < ((size_t)4 << 20) ? 1 : host_threads;
        if (nt <= 1) { work(0, items.size()); return; }
        if (!copy_pool || copy_pool->size() != nt) copy_pool.reset(new CopyPool(nt));
        const std::function<void(size_t, size_t)> fn = work;
        copy_pool->run(items.size(), fn);
    }
    // arrs must be ordered: copied in only, copied both ways, copied back only.  run(stream, nc, c0, device base of the slot) enqueues
    // the solver for one chunk whose arrays lie at dev + a.off, dense with leading dimension nc (slots are sized for
    // cn = min(ncol, host_chunk) columns).  err_dev: the solver's device-side input-assertion word (null: the scheme has none); it
    // comes back with every chunk's outputs, and a chunk whose word is set is NOT scattered into the caller's arrays - the pipeline
    // drains and returns PIPE_FLAGGED for the caller's check() to turn into the reference's message (the reference stops before it
    // computes anything; here the chunks before the offending one have been delivered).
    int host_pipeline(int ncol, std::vector<PipeArr> &arrs, const std::function<int(hipStream_t, int, int, char *, int)> &run,
                      const uint32_t *err_dev = nullptr)
    {
        const int cn = ncol < host_chunk ? ncol : host_chunk;
        const int ld_host = host_ld ? (int)host_ld : ncol;
        // a chunk's arrays lie dense in its slot (leading dimension = the chunk's columns): the offsets are those of the chunk's own size, so
        // that a small chunk moves only its own bytes
        size_t in_end = 0, out_begin = 0, total = 0;
        auto layout = [&](int nc_) {
            size_t off = 0; in_end = 0; out_begin = (size_t)-1;
            for (auto &a : arrs) {
                a.off = off;
                if (a.dst && out_begin == (size_t)-1) out_begin = off;
                off += (a.rows * (size_t)nc_ * a.ebytes + 255) & ~(size_t)255;
                if (a.src) in_end = off;
            }
            if (out_begin == (size_t)-1) out_begin = off;
            total = off;
        };
        layout(cn);
        const size_t total_max = total, in_bytes_max = in_end, out_bytes_max = total - out_begin;
#define PIPECHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(GEOSRAD_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
        if (!pipe_h2d) {
            PIPECHK(hipStreamCreateWithFlags(&pipe_h2d, hipStreamNonBlocking));
            PIPECHK(hipStreamCreateWithFlags(&pipe_d2h, hipStreamNonBlocking));
            for (int s = 0; s < PIPE_SLOTS; s++) for (int e = 0; e < 3; e++) PIPECHK(hipEventCreateWithFlags(&pipe_ev[s][e], hipEventDisableTiming));
            PIPECHK(hipHostMalloc((void **)&pipe_err, PIPE_SLOTS * sizeof(uint32_t), hipHostMallocDefault));
        }
        if (total_max > pipe_dev_bytes || in_bytes_max > pipe_pin_bytes[0] || out_bytes_max > pipe_pin_bytes[1]) {
            PIPECHK(hipDeviceSynchronize());
            const size_t want_dev = total_max > pipe_dev_bytes ? total_max : pipe_dev_bytes;
            const size_t want_pin[2] = {in_bytes_max > pipe_pin_bytes[0] ? in_bytes_max : pipe_pin_bytes[0],
                                        out_bytes_max > pipe_pin_bytes[1] ? out_bytes_max : pipe_pin_bytes[1]};
            pipe_dev_bytes = pipe_pin_bytes[0] = pipe_pin_bytes[1] = 0;      // a failure below leaves "nothing allocated", not stale sizes
            for (int s = 0; s < PIPE_SLOTS; s++) {
                if (pipe_dev[s]) { (void)hipFree(pipe_dev[s]); pipe_dev[s] = nullptr; }
                for (int d = 0; d < 2; d++) if (pipe_pin[s][d]) { (void)hipHostFree(pipe_pin[s][d]); pipe_pin[s][d] = nullptr; }
            }
            for (int s = 0; s < PIPE_SLOTS; s++) {
                if (hipMalloc((void **)&pipe_dev[s], want_dev) != hipSuccess) { pipe_release(); return fail(GEOSRAD_ENOMEM, "hipMalloc of the host-API staging slot failed"); }
                for (int d = 0; d < 2; d++)
                    if (hipHostMalloc((void **)&pipe_pin[s][d], want_pin[d] ? want_pin[d] : 256, hipHostMallocDefault) != hipSuccess) {
                        pipe_release();
                        return fail(GEOSRAD_ENOMEM, "hipHostMalloc of the pinned host-API staging slot failed");
                    }
            }
            pipe_dev_bytes = want_dev; pipe_pin_bytes[0] = want_pin[0]; pipe_pin_bytes[1] = want_pin[1];
        }
        for (int s = 0; s < PIPE_SLOTS; s++) pipe_err[s] = 0;
        bool input_error = false;
        // chunk boundaries.  (Small first chunks that double up to cn - so that the first transfer starts before 171 / 265 MB have been
        // gathered - were measured: 58.3 against 56.7 ms per RRTMG_LW + RRTMG_SW call pair of 97 200 columns; the call is bound by the
        // H2D transfer itself, 2.59 GB at the box's 57.5 GB/s = 45 ms, profiles/r04_host_api.md.)
        std::vector<int> cstart;
        for (int c = 0; c < ncol; c += cn) cstart.push_back(c);
        cstart.push_back(ncol);
        const int nchunks = (int)cstart.size() - 1;
        const bool trace = getenv("GEOSRAD_HOST_TRACE") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_g = 0, t_s = 0, t_w = 0, t_e = 0;
        const double t_begin = now();
        for (int k = 0; k < nchunks + PIPE_LAG; k++) {
            if (k < nchunks) {
                const int s = k % PIPE_SLOTS, c0 = cstart[k], nc = cstart[k + 1] - c0;
                double t0 = now();
                if (k >= PIPE_SLOTS) PIPECHK(hipEventSynchronize(pipe_ev[s][0]));   // the slot's previous transfer has left the staging memory
                double t1 = now(); t_w += t1 - t0;
                layout(nc);
                const size_t in_bytes = in_end, out_bytes = total - out_begin;
                pipe_copy(arrs, pipe_pin[s][0], 0, ld_host, c0, nc, true);
                t0 = now(); t_g += t0 - t1;
                if (k >= PIPE_SLOTS) PIPECHK(hipStreamWaitEvent(pipe_h2d, pipe_ev[s][2], 0));   // ... and its previous chunk has been copied out of the device slot
                if (in_bytes) PIPECHK(hipMemcpyAsync(pipe_dev[s], pipe_pin[s][0], in_bytes, hipMemcpyHostToDevice, pipe_h2d));
                PIPECHK(hipEventRecord(pipe_ev[s][0], pipe_h2d));
                PIPECHK(hipStreamWaitEvent(stream, pipe_ev[s][0], 0));
                const int rc = run(stream, nc, c0, pipe_dev[s], cn);
                if (rc) { (void)hipDeviceSynchronize(); return rc; }
                PIPECHK(hipEventRecord(pipe_ev[s][1], stream));
                PIPECHK(hipStreamWaitEvent(pipe_d2h, pipe_ev[s][1], 0));
                if (out_bytes) PIPECHK(hipMemcpyAsync(pipe_pin[s][1], pipe_dev[s] + out_begin, out_bytes, hipMemcpyDeviceToHost, pipe_d2h));
                if (err_dev) PIPECHK(hipMemcpyAsync(&pipe_err[s], err_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, pipe_d2h));
                PIPECHK(hipEventRecord(pipe_ev[s][2], pipe_d2h));
                t_e += now() - t0;
            }
            if (k >= PIPE_LAG && k - PIPE_LAG < nchunks) {
                const int j = k - PIPE_LAG, s = j % PIPE_SLOTS, c0 = cstart[j], nc = cstart[j + 1] - c0;
                double t0 = now();
                PIPECHK(hipEventSynchronize(pipe_ev[s][2]));
                double t1 = now(); t_w += t1 - t0;
                if (err_dev && pipe_err[s]) { input_error = true; break; }      // this chunk (or one enqueued behind it) tripped an input assertion
                layout(nc);
                pipe_copy(arrs, pipe_pin[s][1], out_begin, ld_host, c0, nc, false);
                t_s += now() - t1;
            }
        }
        if (input_error) { PIPECHK(hipDeviceSynchronize()); return PIPE_FLAGGED; }
        if (trace)
            fprintf(stderr, "geosrad host pipeline: %d columns, %d chunks of %d, %.1f MB in / %.1f MB out per chunk: total %.1f ms = gather %.1f + "
                            "scatter %.1f + enqueue %.1f + waiting for the GPU %.1f\n", ncol, nchunks, cn, in_bytes_max / 1e6, out_bytes_max / 1e6,
                    now() - t_begin, t_g, t_s, t_e, t_w);
#undef PIPECHK
        return GEOSRAD_OK;
    }
    virtual ~geosrad_ctx() { pipe_release(); }
    virtual int init() = 0;
    virtual int set_tables_lw(const void *blob, size_t n) = 0;
    virtual int set_inhomogeneity(int ih, const void *blob, size_t n) = 0;
    virtual int set_corr(const double *adl, const double *rdl) = 0;
    virtual size_t workspace_bytes() const = 0;
    // RATS diagnostics of LW_Driver (GEOS_IrradGridComp.F90:3405-3468): total-sky profiles with one gas removed, per gas
    struct LwRats { int n; int gas[GEOSRAD_RAT_NGAS]; void *uflx, *dflx, *duflx_dTs; };      // outputs [n][nlay+1][ncol]
    virtual int lw_dev(hipStream_t st, int ncol, int nlay, int dudTs, const void *const *in, int iceflg, int liqflg, int dyofyr,
                       int cloudLM, int cloudMH, int32_t *clearCounts, void *const *out, const int32_t *band_output,
                       void *dbg_taug, void *dbg_pfracs, const LwRats *rats) = 0;
    virtual int lw_host(int ncol, int nlay, int dudTs, const void *const *in, int iceflg, int liqflg, int dyofyr, int cloudLM,
                        int cloudMH, int32_t *clearCounts, void *const *out, const int32_t *band_output, void *taug,
                        void *pfracs) = 0;
    virtual int mcica_host(int ncol, int nsubcol, int nlay, const void *zmid, const void *alat, int doy, const void *play,
                           const void *cldfrac, const void *ciwp, const void *clwp, double cwp_tiny, const int32_t *so,
                           int32_t *cldy, void *ciwp_s, void *clwp_s) = 0;
    virtual int mcica_dev(hipStream_t st, int ncol, int nsubcol, int nlay, const void *zmid, const void *alat, int doy, const void *play,
                          const void *cldfrac, const void *ciwp, const void *clwp, double cwp_tiny, const int32_t *so,
                          int32_t *cldy, void *ciwp_s, void *clwp_s) = 0;
    virtual int check(hipStream_t st, int which = -1) = 0;      // which: -1 either solver's assertions, 0 RRTMG_LW (+ McICA), 1 RRTMG_SW
    virtual int set_tables_sw(const void *blob, size_t n) = 0;
    virtual int set_tables_chou_lw(const void *blob, size_t n) = 0;
    virtual int set_tables_chou_sw(const void *blob, size_t n) = 0;
    virtual int sorad_dev(hipStream_t st, int m, int np, int nb, const void *const *in, double co2, int ict, int icb, const void *hk_uv,
                          const void *hk_ir, void *const *out, int do_drfband) = 0;
    virtual int sorad_host(int m, int np, int nb, const void *const *in, double co2, int ict, int icb, const void *hk_uv, const void *hk_ir,
                           void *const *out, int do_drfband) = 0;
    virtual int irrad_dev(hipStream_t st, int m, int np, const void *const *in, double co2, int trace, int ict, int icb, int ns, int na,
                          int nb, void *const *aer, void *const *out) = 0;
    virtual int irrad_host(int m, int np, const void *const *in, double co2, int trace, int ict, int icb, int ns, int na, int nb,
                           void *const *aer, void *const *out) = 0;
    virtual int sw_dev(hipStream_t st, int ncol, int nlay, double scon, double adjes, int isolvar, const void *const *in, int iceflg,
                       int liqflg, int dyofyr, int iaer, int cloudLM, int cloudMH, int normFlx, int32_t *clearCounts, void *const *out,
                       int do_drfband, const void *bndscl, const void *indsolvar, const void *solcycfrac,
                       void *const *dbg) = 0;
    virtual int lw_driver_dev(hipStream_t st, int ncol, int lm, int nb, const void *const *in, const double *consts, int iceflg,
                              int liqflg, int doy, int lcldlm, int lcldmh, const int32_t *band_output, void *const *out, int nrats,
                              const int32_t *rat_gas, void *const *rat_out) = 0;
    virtual int sw_driver_dev(hipStream_t st, int ncol, int lm, int nb, const void *const *in, const double *consts, int iceflg,
                              int liqflg, double sc, double dist, int isolvar, int dyofyr, int include_aerosols, int lcldlm,
                              int lcldmh, int normflx, const void *bndsolvar, const void *indsolvar, void *const *out) = 0;
    virtual int lw_chou_post_dev(hipStream_t st, int ncol, int lm, const void *const *in, void *const *out) = 0;
    virtual int sw_driver_chou_dev(hipStream_t st, int ncol, int lm, const void *const *in, const double *consts, int lcldmh, int lcldlm,
                                   const void *hk_uv, const void *hk_ir, int do_drfband, void *const *out) = 0;
    virtual int lw_update_flx_dev(hipStream_t st, int ncol, int lm, int rrtmg, int lev_mid_high, int lev_low_mid, double undef,
                                  const void *const *in, void *const *out) = 0;
    virtual int lw_update_rats_dev(hipStream_t st, int ncol, int lm, int nrats, const void *const *in, void *const *out) = 0;
    virtual int lw_update_bands_dev(hipStream_t st, int ncol, const int32_t *band_output, const double *wn1, const double *wn2, double undef,
                                    const void *tsinst, const void *ts_int, const void *olrb_int, const void *dolrb_int, void *olrb_exp,
                                    void *tbrb_exp) = 0;
    virtual int sw_update_export_dev(hipStream_t st, int ncol, int lm, int nbands, const void *const *in, void *const *out) = 0;
    virtual int sw_update_surface_dev(hipStream_t st, int ncol, int lm, double undef, const void *const *in, void *const *out) = 0;
    virtual int rad_tendencies_dev(hipStream_t st, int ncol, int lm, double grav, double cp, const void *const *in,
                                   void *const *out) = 0;
    virtual int sw_host(int ncol, int nlay, double scon, double adjes, int isolvar, const void *const *in, int iceflg, int liqflg,
                        int dyofyr, int iaer, int cloudLM, int cloudMH, int normFlx, int32_t *clearCounts, void *const *out,
                        int do_drfband, const void *bndscl, const void *indsolvar, const void *solcycfrac,
                       void *const *dbg) = 0;
    virtual int lit_index_dev(hipStream_t st, int ncol, const void *zth, int32_t *idx, int32_t *pos, int32_t *nlit_dev, int *nlit_host) = 0;
    virtual int lit_pack_dev(hipStream_t st, int pdim, int udim, int nlev, const int32_t *idx, const int32_t *nlit_dev, const void *unpacked,
                             void *packed) = 0;
    virtual int lit_unpack_dev(hipStream_t st, int pdim, int udim, int nlev, const int32_t *pos, const void *packed, void *unpacked,
                               int use_default, double dflt) = 0;
};

// order of the `in` / `out` pointer arrays of sw_dev / sw_host
enum SwIn { S_PLAY, S_PLEV, S_TLAY, S_H2O, S_O3, S_CO2, S_CH4, S_O2, S_CLD, S_CIWP, S_CLWP, S_REI, S_REL, S_ZM, S_ALAT, S_TAUAER,
            S_SSAAER, S_ASMAER, S_COSZEN, S_ASDIR, S_ASDIF, S_ALDIR, S_ALDIF, S_NIN };
enum SwOutIx { SO_UFLX, SO_DFLX, SO_UFLXC, SO_DFLXC, SO_NIRR, SO_NIRF, SO_PARR, SO_PARF, SO_UVRR, SO_UVRF, SO_FSWBAND, SO_COT0,
               SO_DRBAND = SO_COT0 + 8, SO_DFBAND, SO_NOUT };

// sorad: order of the `in` (15) / `out` (13) pointer arrays
enum SoIn { SI_COSZ, SI_PL, SI_TA, SI_WA, SI_OA, SI_CWC, SI_FCLD, SI_REFF, SI_TAUA, SI_SSAA, SI_ASYA, SI_RSUVBM, SI_RSUVDF, SI_RSIRBM,
            SI_RSIRDF, SI_NIN };
enum SoOutIx { SOO_FLX, SOO_FLC, SOO_FDIRUV, SOO_FDIFUV, SOO_FDIRPAR, SOO_FDIFPAR, SOO_FDIRIR, SOO_FDIFIR, SOO_FLXU, SOO_FLCU,
               SOO_SFCBAND, SOO_DRBAND, SOO_DFBAND, SOO_NOUT };

// irrad: order of the `in` (19) / `aer` (3, in-out) / `out` (11) pointer arrays
enum ChIn { C_PLE, C_TA, C_WA, C_OA, C_TB, C_N2O, C_CH4, C_CFC11, C_CFC12, C_CFC22, C_CWC, C_FCLD, C_REFF, C_FS, C_TG, C_EG, C_TV, C_EV,
            C_RV, C_NIN };
enum ChOutIx { CO_FLXU, CO_FLCU, CO_FLAU, CO_FLXAU, CO_FLXD, CO_FLCD, CO_FLAD, CO_FLXAD, CO_DFDTS, CO_SFCEM, CO_TAUDIAG, CO_NOUT };

// order of the `in` pointer array of lw_dev / lw_host
enum LwIn { I_PLAY, I_PLEV, I_TLAY, I_TLEV, I_TSFC, I_EMIS, I_H2O, I_O3, I_CO2, I_CH4, I_N2O, I_O2, I_CFC11, I_CFC12, I_CFC22,
            I_CCL4, I_CLDF, I_CIWP, I_CLWP, I_REI, I_REL, I_TAUAER, I_ZM, I_ALAT, I_NIN };
enum LwOutIx { O_UFLX, O_DFLX, O_UFLXC, O_DFLXC, O_DUFLX, O_DUFLXC, O_OLRB, O_DOLRB, O_NOUT };

// The library is built from this one source as three objects compiled in parallel (GEOSRAD_PART = 4: the fp32
// instantiation of Ctx and of every kernel, 8: the fp64 one, 0: the extern "C" layer); without GEOSRAD_PART it is
// a single translation unit.
geosrad_ctx *geosrad_new_ctx_f32();
geosrad_ctx *geosrad_new_ctx_f64();

#if !defined(GEOSRAD_PART) || GEOSRAD_PART != 0
namespace {

template <typename R> struct Ctx : geosrad_ctx {
    using R2 = typename Vec2<R>::T;
    // tables
    char *d_tab = nullptr; size_t tab_bytes = 0;
    char *d_xcw = nullptr; size_t xcw_bytes = 0;
    size_t so_lds_set = 0;       // dynamic-LDS limit granted to k_sorad_col so far
    LwDev<R> h_T{};            // host copy (device pointers inside)
    LwDev<R> *d_T = nullptr;
    bool have_lw = false;
    // RRTMG_SW tables
    char *d_tab_sw = nullptr; size_t tab_sw_bytes = 0;
    SwDev<R> h_S{};
    std::vector<R> avgcyc_mg, avgcyc_sb;          // NRLSSI2 mgavgcyc / sbavgcyc (134 each), host only: isolvar == 1
    SwDev<R> *d_S = nullptr;
    bool have_sw = false;
    char *d_ws_sw = nullptr; size_t ws_sw_bytes = 0; int ws_sw_ncol = 0, ws_sw_nlay = 0, ws_sw_planes = 0;
    // Chou-Suarez SW tables + workspace
    char *d_tab_so = nullptr; size_t tab_so_bytes = 0;
    SoradDev<R> h_O{};
    SoradDev<R> *d_O = nullptr;
    bool have_sorad = false;
    char *d_ws_so = nullptr; size_t ws_so_bytes = 0;
    char *d_ws_swc = nullptr; size_t ws_swc_bytes = 0;      // sw_driver_chou_dev: the arrays SORADCORE prepares for sorad
    // Chou-Suarez LW tables + workspace
    char *d_tab_ch = nullptr; size_t tab_ch_bytes = 0;
    ChouDev<R> h_C{};
    ChouDev<R> *d_C = nullptr;
    bool have_chou = false;
    char *d_ws_ch = nullptr; size_t ws_ch_bytes = 0;
    char *d_ws_drvs[2] = {nullptr, nullptr}; size_t ws_drvs_bytes[2] = {0, 0};      // RRTMG-side arrays of the LW / SW GridComp drivers (separate: the two may run on two streams)
    // McICA segment plans (jump-ahead constants), cached per (mode, nsubcol, nlay, inhomogeneous?)
    struct PlanEntry { McSegDev *d_seg; int nseg; KissJump jsub, jhalf; };
    std::map<std::tuple<int, int, int, int>, PlanEntry> plans;
    // workspace
    char *d_ws = nullptr; size_t ws_bytes = 0; int ws_ncol = 0, ws_nlay = 0;
    char *d_zero = nullptr; size_t zero_bytes = 0;      // all-zero (nlay, ncol) plane of the RATS passes
    int *d_bandflags = nullptr;                         // Update_Flx band exports: "band has a non-zero flux somewhere"
    uint32_t *d_err = nullptr;
    // staging for host-pointer entry points
    char *d_io = nullptr; size_t io_bytes = 0;

    Ctx() { for (int i = 0; i < 4; i++) { h_T.aam[i] = 0; h_T.ram[i] = 0; } }
    ~Ctx() override
    {
        if (d_tab) (void)hipFree(d_tab);
        if (d_xcw) (void)hipFree(d_xcw);
        if (d_T) (void)hipFree(d_T);
        if (d_ws) (void)hipFree(d_ws);
        if (d_zero) (void)hipFree(d_zero);
        if (d_bandflags) (void)hipFree(d_bandflags);
        for (auto &pe : plans) if (pe.second.d_seg) (void)hipFree(pe.second.d_seg);
        if (d_tab_sw) (void)hipFree(d_tab_sw);
        if (d_tab_ch) (void)hipFree(d_tab_ch);
        if (d_tab_so) (void)hipFree(d_tab_so);
        if (d_O) (void)hipFree(d_O);
        if (d_ws_so) (void)hipFree(d_ws_so);
        if (d_ws_swc) (void)hipFree(d_ws_swc);
        if (d_C) (void)hipFree(d_C);
        if (d_ws_ch) (void)hipFree(d_ws_ch);
        for (char *q : d_ws_drvs) if (q) (void)hipFree(q);
        if (d_S) (void)hipFree(d_S);
        if (d_ws_sw) (void)hipFree(d_ws_sw);
        if (d_err) (void)hipFree(d_err);
        if (d_io) (void)hipFree(d_io);
        if (d_mc) (void)hipFree(d_mc);
        for (auto &e : sa_jumps) if (e.second) (void)hipFree(e.second);
        if (stream) (void)hipStreamDestroy(stream);
    }

    int init() override
    {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipMalloc((void **)&d_err, 256));
        HIPCHK(hipMemset(d_err, 0, 256));
        HIPCHK(hipMalloc((void **)&d_T, sizeof(LwDev<R>)));
        HIPCHK(hipMalloc((void **)&d_S, sizeof(SwDev<R>)));
        HIPCHK(hipMalloc((void **)&d_C, sizeof(ChouDev<R>)));
        HIPCHK(hipMalloc((void **)&d_O, sizeof(SoradDev<R>)));
        if (lw_bands_lds_bytes<R>() > 64 * 1024) {      // the LDS copy of the LW transmittance table (fp32 build)
            const int lds = (int)lw_bands_lds_bytes<R>();
            HIPCHK(hipFuncSetAttribute((const void *)k_lw_bands<R, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            if constexpr (sizeof(R) == 4)
                HIPCHK(hipFuncSetAttribute((const void *)k_lw_bands<R, false, false, LW_WIDE_BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIPCHK(hipFuncSetAttribute((const void *)k_lw_bands<R, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIPCHK(hipFuncSetAttribute((const void *)k_lw_bands<R, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        }
        // Oreopoulos et al. (2012) defaults (cloud_subcol_gen.F90:51-59)
        const double adl[4] = {1.4315, 2.1219, 7., -25.584}, rdl[4] = {0.72192, 0.78996, 8.5, 40.404};
        for (int i = 0; i < 4; i++) { h_T.aam[i] = (R)(sizeof(R) == 4 ? (float)adl[i] : adl[i]); h_T.ram[i] = (R)(sizeof(R) == 4 ? (float)rdl[i] : rdl[i]); }
        // the McICA kernels of the SW solver and of the stand-alone generator read aam / ram / xcw from d_T: it must hold the
        // defaults (and a null xcw) even when neither the LW tables, an inhomogeneity table nor correlation lengths are ever set
        return sync_T();
    }

    int sync_T()
    {
        HIPCHK(hipMemcpy(d_T, &h_T, sizeof(LwDev<R>), hipMemcpyHostToDevice));
        return GEOSRAD_OK;
    }

    // ---- table upload ----------------------------------------------------------------------------------
    int set_tables_lw(const void *blob, size_t nbytes) override
    {
        HIPCHK(hipSetDevice(device));
        Blob B;
        if (!B.parse(blob, nbytes)) return fail(GEOSRAD_ETABLE, B.err);
        if (B.realbytes != (int)sizeof(R))
            return fail(GEOSRAD_ETABLE, "table blob real size does not match the context's real_kind (use the _r4 blob for "
                                        "real_kind 4 and the _r8 blob for real_kind 8)");
        TableStage<R> S(B);
        std::string &missing = S.missing;
        auto &fix = S.fix;
        auto get = [&](const std::string &nm, size_t expect) { return S.get(nm, expect); };
        auto reserve = [&](size_t nreal) { return S.reserve(nreal); };
        auto at = [&](size_t off) { return S.at(off); };
        auto tr2 = [&](const R **slot, const std::string &nm, int n1, int ng) { S.tr2(slot, nm, n1, ng); };
        auto tr3 = [&](const R **slot, const std::string &nm, int nsp, int ng) { S.tr3(slot, nm, nsp, ng); };
        auto rows = [&](const R **slot, const std::string &nm, int ng, int m) { S.rows(slot, nm, ng, m); };
        auto raw = [&](const R **slot, const std::string &nm, size_t cnt) { S.raw(slot, nm, cnt); };
        auto scalar = [&](const std::string &nm) { return S.scalar(nm); };

        static const int ng[17] = {0, 10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2};
        static const int nspa[17] = {0, 1, 1, 9, 9, 9, 1, 9, 1, 9, 1, 1, 9, 9, 1, 9, 9};
        static const int nspb[17] = {0, 1, 1, 5, 5, 5, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0, 0};
        LwDev<R> &T = h_T;
        const R *xcw_keep = T.xcw;
        R aam[4], ram[4];
        for (int i = 0; i < 4; i++) { aam[i] = T.aam[i]; ram[i] = T.ram[i]; }
        memset(&T, 0, sizeof(T));
        T.xcw = xcw_keep;
        for (int i = 0; i < 4; i++) { T.aam[i] = aam[i]; T.ram[i] = ram[i]; }
        char nm[64];
        for (int b = 1; b <= 16; b++) {
            BandTab<R> &bt = T.b[b];
            auto N = [&](const char *s) { snprintf(nm, sizeof nm, "b%02d_%s", b, s); return std::string(nm); };
            tr2(&bt.absa, N("absa"), 65 * nspa[b], ng[b]);
            // band 16 declares absb(235,ng) although nspb(16) = 0 (rrlw_kg16.F90); bands 6,12,13,15 have no absb
            if (nspb[b] > 0 || b == 16) tr2(&bt.absb, N("absb"), 235 * (b == 16 ? 1 : nspb[b]), ng[b]);
            rows(&bt.fracrefa, N("fracrefa"), ng[b], nspa[b] == 9 ? 9 : 1);
            if (b != 6 && b != 12 && b != 15) rows(&bt.fracrefb, N("fracrefb"), ng[b], nspb[b] == 5 ? 5 : 1);
            tr2(&bt.selfref, N("selfref"), 10, ng[b]);
            tr2(&bt.forref, N("forref"), 4, ng[b]);
        }
        tr2(&T.b[1].m[0], "b01_ka_mn2", 19, ng[1]);   tr2(&T.b[1].m[1], "b01_kb_mn2", 19, ng[1]);
        tr3(&T.b[3].m[0], "b03_ka_mn2o", 9, ng[3]);   tr3(&T.b[3].m[1], "b03_kb_mn2o", 5, ng[3]);
        tr3(&T.b[5].m[0], "b05_ka_mo3", 9, ng[5]);    rows(&T.b[5].m[1], "b05_ccl4", ng[5], 1);
        tr2(&T.b[6].m[0], "b06_ka_mco2", 19, ng[6]);  rows(&T.b[6].m[1], "b06_cfc11adj", ng[6], 1);
        rows(&T.b[6].m[2], "b06_cfc12", ng[6], 1);
        tr3(&T.b[7].m[0], "b07_ka_mco2", 9, ng[7]);   tr2(&T.b[7].m[1], "b07_kb_mco2", 19, ng[7]);
        tr2(&T.b[8].m[0], "b08_ka_mco2", 19, ng[8]);  tr2(&T.b[8].m[1], "b08_kb_mco2", 19, ng[8]);
        tr2(&T.b[8].m[2], "b08_ka_mo3", 19, ng[8]);   tr2(&T.b[8].m[3], "b08_ka_mn2o", 19, ng[8]);
        tr2(&T.b[8].m[4], "b08_kb_mn2o", 19, ng[8]);  rows(&T.b[8].m[5], "b08_cfc12", ng[8], 1);
        rows(&T.b[8].m[6], "b08_cfc22adj", ng[8], 1);
        tr3(&T.b[9].m[0], "b09_ka_mn2o", 9, ng[9]);   tr2(&T.b[9].m[1], "b09_kb_mn2o", 19, ng[9]);
        tr2(&T.b[11].m[0], "b11_ka_mo2", 19, ng[11]); tr2(&T.b[11].m[1], "b11_kb_mo2", 19, ng[11]);
        tr3(&T.b[13].m[0], "b13_ka_mco2", 9, ng[13]); tr3(&T.b[13].m[1], "b13_ka_mco", 9, ng[13]);
        tr2(&T.b[13].m[2], "b13_kb_mo3", 19, ng[13]);
        tr3(&T.b[15].m[0], "b15_ka_mn2", 9, ng[15]);

        raw(&T.totplnk, "totplnk", 181 * 16); raw(&T.totplnkderiv, "totplnkderiv", 181 * 16);
        raw(&T.preflog, "preflog", 59); raw(&T.tref, "tref", 59); raw(&T.chi_mls, "chi_mls", 7 * 59);
        raw(&T.tau_tbl, "tau_tbl", NTBL + 1);
        raw(&T.absice0, "absice0", 2); raw(&T.absice1, "absice1", 10); raw(&T.absice2, "absice2", 43 * 16);
        raw(&T.absice3, "absice3", 46 * 16); raw(&T.absice4, "absice4", 200 * 16); raw(&T.absliq1, "absliq1", 58 * 16);
        {   // interleaved (exp_tbl, tfn_tbl)
            const R *ex = get("exp_tbl", NTBL + 1), *tf = get("tfn_tbl", NTBL + 1);
            if (ex && tf) {
                size_t off = reserve(2 * (size_t)(NTBL + 1));
                for (int i = 0; i <= NTBL; i++) { at(off)[2 * i] = ex[i]; at(off)[2 * i + 1] = tf[i]; }
                fix.push_back({(const R **)&T.lut, off});
            }
        }
        {   // chi_mls ratio tables: the same IEEE divisions setcoef performs per layer
            const R *chi = get("chi_mls", 7 * 59);
            if (chi) {
                static const int pa[RAT_NPAIR] = {1, 1, 1, 1, 4, 3}, pb[RAT_NPAIR] = {2, 3, 4, 6, 2, 2};
                size_t off = reserve((size_t)RAT_NPAIR * 60);
                for (int p = 0; p < RAT_NPAIR; p++)
                    for (int j = 1; j <= 59; j++) at(off)[p * 60 + j] = chi[(j - 1) * 7 + pa[p] - 1] / chi[(j - 1) * 7 + pb[p] - 1];
                fix.push_back({&T.rat, off});
            }
        }
        T.bpade = scalar("bpade"); T.fluxfac = scalar("fluxfac"); T.oneminus = scalar("oneminus");
        T.grav = scalar("grav"); T.avogad = scalar("avogad");
        {
            const R *dw = get("delwave", 16);
            if (dw) for (int b = 1; b <= 16; b++) T.delwave[b] = dw[b - 1];
            auto it = B.e.find("ice1b");
            if (it == B.e.end() || it->second.kind != -4 || it->second.count != 16) missing += "ice1b ";
            else memcpy(T.ice1b, it->second.data, 16 * sizeof(int32_t));
            // the band <-> g-point map is compiled into the kernels; refuse tables that disagree
            auto ig = B.e.find("ngb");
            if (ig == B.e.end() || ig->second.count != 140) missing += "ngb ";
            else {
                const int32_t *ngb = (const int32_t *)ig->second.data;
                int g = 0;
                for (int b = 1; b <= 16; b++) for (int k = 0; k < ng[b]; k++, g++) if (ngb[g] != b) missing += "ngb(mismatch) ";
            }
        }
        if (!missing.empty()) return fail(GEOSRAD_ETABLE, "missing/ill-shaped table entries: " + missing);

        if (d_tab) { HIPCHK(hipFree(d_tab)); d_tab = nullptr; }
        tab_bytes = S.stage.size();
        HIPCHK(hipMalloc((void **)&d_tab, tab_bytes));
        HIPCHK(hipMemcpy(d_tab, S.stage.data(), tab_bytes, hipMemcpyHostToDevice));
        for (auto &f : fix) *f.first = (const R *)(d_tab + f.second);
        have_lw = true;
        return sync_T();
    }

    int set_inhomogeneity(int ih, const void *blob, size_t nbytes) override
    {
        HIPCHK(hipSetDevice(device));
        if (ih == 0) {
            h_T.xcw = nullptr;
            return sync_T();
        }
        if (ih != 1 && ih != 2) return fail(GEOSRAD_EINPUT, "unknown inhomogeneity type");
        Blob B;
        if (!blob || !B.parse(blob, nbytes)) return fail(GEOSRAD_ETABLE, blob ? B.err : "xcw blob required for ih > 0");
        auto it = B.e.find("xcw");
        if (B.realbytes != (int)sizeof(R) || it == B.e.end() || it->second.count != 140000 || it->second.kind != (int)sizeof(R))
            return fail(GEOSRAD_ETABLE, "xcw blob missing / wrong precision");
        auto ii = B.e.find("ih");
        if (ii != B.e.end() && *(const int32_t *)ii->second.data != ih) return fail(GEOSRAD_ETABLE, "xcw blob is for a different ih");
        if (!d_xcw) HIPCHK(hipMalloc((void **)&d_xcw, 140000 * sizeof(R)));
        HIPCHK(hipMemcpy(d_xcw, it->second.data, 140000 * sizeof(R), hipMemcpyHostToDevice));
        h_T.xcw = (const R *)d_xcw;
        return sync_T();
    }

    int set_corr(const double *adl, const double *rdl) override
    {
        // parameters are default-real in the reference: round through float for real_kind 4
        for (int i = 0; i < 4; i++) {
            if (adl) h_T.aam[i] = (R)adl[i];
            if (rdl) h_T.ram[i] = (R)rdl[i];
        }
        return sync_T();
    }

    size_t workspace_bytes() const override { return ws_bytes + ws_sw_bytes + ws_ch_bytes + ws_so_bytes + ws_drvs_bytes[0] + ws_drvs_bytes[1] + io_bytes + tab_bytes + tab_sw_bytes + tab_ch_bytes; }

    // ---- workspace -------------------------------------------------------------------------------------
    struct Ws { R *sc; uint32_t *scidx; R *pwvcm; uint8_t *colcloudy, *laycloudy; int32_t *perm, *nclear; R *taucmc, *alpha, *rcorr; uint16_t *s1, *s2; R *part;
                uint32_t *pfcode; R *pffs; };
    static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
    size_t ws_layout(int nc, int nlay, Ws *w, char *base) const
    {
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return base ? base + o : (char *)nullptr; };
        const size_t cl = (size_t)nlay * nc;
        char *p;
        p = take(SC_NFIELD * cl * sizeof(R)); if (w) w->sc = (R *)p;
        p = take(cl * 4); if (w) w->scidx = (uint32_t *)p;
        p = take((size_t)nc * sizeof(R)); if (w) w->pwvcm = (R *)p;
        p = take(nc); if (w) w->colcloudy = (uint8_t *)p;
        p = take((size_t)nc * 4); if (w) w->perm = (int32_t *)p;
        p = take(4); if (w) w->nclear = (int32_t *)p;
        p = take(cl); if (w) w->laycloudy = (uint8_t *)p;
        p = take(cl * sizeof(R)); if (w) w->alpha = (R *)p;
        p = take(cl * sizeof(R)); if (w) w->rcorr = (R *)p;
        p = take(NG_LW * cl * sizeof(R)); if (w) w->taucmc = (R *)p;
        const size_t clp = (size_t)nlay * (((size_t)nc + 255) & ~(size_t)255);      // tiled by 256-column block
        // parked cells of the band sweeps: a 2-byte Pade index per (layer, g-point) and stream (lw_kernels.hpp band_body)
        p = take(NG_LW * clp * sizeof(uint16_t)); if (w) w->s1 = (uint16_t *)p;
        p = take(NG_LW * clp * sizeof(uint16_t)); if (w) w->s2 = (uint16_t *)p;
        p = take((size_t)6 * NB_LW * (nlay + 1) * nc * sizeof(R)); if (w) w->part = (R *)p;
        // split path: the Planck-fraction selector of every (band, layer, column) (lw_split_kernels.hpp)
        p = take(lw_split_path ? NB_LW * cl * 4 : 0); if (w) w->pfcode = (uint32_t *)p;
        p = take(lw_split_path ? NB_LW * cl * sizeof(R) : 0); if (w) w->pffs = (R *)p;
        return off;
    }
    int ensure_ws(int nc, int nlay)
    {
        if (d_ws && nc <= ws_ncol && nlay == ws_nlay) return GEOSRAD_OK;
        // grow-only in columns for a given nlay
        const int want = (d_ws && nlay == ws_nlay && nc < ws_ncol) ? ws_ncol : nc;
        if (d_ws) { HIPCHK(hipFree(d_ws)); d_ws = nullptr; ws_bytes = 0; }
        const size_t need = ws_layout(want, nlay, nullptr, nullptr);
        hipError_t e = hipMalloc((void **)&d_ws, need);
        if (e != hipSuccess) return fail(GEOSRAD_ENOMEM, "hipMalloc of the LW workspace failed (" + std::to_string(need >> 20) +
                                                         " MiB); lower it with geosrad_set_chunk()");
        ws_bytes = need; ws_ncol = want; ws_nlay = nlay;
        return GEOSRAD_OK;
    }


    // quads of sub-columns that never straddle a band; jump distances in units of draws
    int mc_plan(int mode, int nsubcol, int nlay, McPlan &out, int &nseg_out)
    {
        const bool inhomo = h_T.xcw != nullptr;
        const auto key = std::make_tuple(mode, nsubcol, nlay, inhomo ? 1 : 0);
        auto it = plans.find(key);
        if (it == plans.end()) {
            std::vector<McSegDev> segs;
            const uint64_t per = (uint64_t)(inhomo ? 4 : 2) * (uint64_t)nlay;
            auto add_range = [&](int g0, int ng, int band) {
                for (int q = 0; q < ng; q += MC_S) {
                    McSegDev sd;
                    memset(&sd, 0, sizeof sd);
                    sd.start = g0 + q; sd.count = (ng - q) < MC_S ? (ng - q) : MC_S; sd.band = band;
                    sd.j = make_kiss_jump((uint64_t)sd.start * per);
                    segs.push_back(sd);
                }
            };
            if (mode == 0) for (int b = 1; b <= NB_LW; b++) add_range(lw_band_g0(b), lw_band_ng(b), b);
            else if (mode == 2) for (int b = 16; b <= 29; b++) add_range(sw_band_g0(b), sw_band_ng(b), b);
            else add_range(0, nsubcol, 0);
            PlanEntry pe;
            pe.nseg = (int)segs.size();
            pe.jsub = make_kiss_jump(per);
            pe.jhalf = make_kiss_jump(2ull * (uint64_t)nlay);
            pe.d_seg = nullptr;
            HIPCHK(hipMalloc((void **)&pe.d_seg, segs.size() * sizeof(McSegDev)));
            HIPCHK(hipMemcpy(pe.d_seg, segs.data(), segs.size() * sizeof(McSegDev), hipMemcpyHostToDevice));
            it = plans.emplace(key, pe).first;
        }
        out.seg = it->second.d_seg; out.nseg = it->second.nseg; out.jsub = it->second.jsub; out.jhalf = it->second.jhalf;
        nseg_out = it->second.nseg;
        return GEOSRAD_OK;
    }

    // ---- RRTMG_LW, device pointers ---------------------------------------------------------------------------
    int lw_dev(hipStream_t st, int ncol, int nlay, int dudTs, const void *const *in, int iceflg, int liqflg, int dyofyr,
               int cloudLM, int cloudMH, int32_t *clearCounts, void *const *out, const int32_t *band_output, void *dbg_taug,
               void *dbg_pfracs, const LwRats *rats) override
    {
        HIPCHK(hipSetDevice(device));
        if (!have_lw) return fail(GEOSRAD_EINVAL, "RRTMG_LW tables not set: call geosrad_set_tables_lw first (rrtmg_lw_ini)");
        if (ncol <= 0 || nlay < 4 || nlay > 203) return fail(GEOSRAD_EINVAL, "bad ncol/nlay (4 <= nlay <= mxlay = 203)");
        // checks the reference performs on scalars
        if (iceflg < 0 || iceflg > 4) return fail(GEOSRAD_EINPUT, "cldprmc: invalid iceflag");
        if (liqflg != 1) return fail(GEOSRAD_EINPUT, "cldprmc: invalid liqflag");
        if (cloudLM == cloudMH) return fail(GEOSRAD_EINPUT, "invalid pressure super-layers!");
        for (int k = 0; k < I_NIN; k++)
            if (!in[k] && k != I_TAUAER) return fail(GEOSRAD_EINVAL, "null input array");
        for (int k = 0; k < 4; k++) if (!out[k]) return fail(GEOSRAD_EINVAL, "null output array");
        if (dudTs && (!out[O_DUFLX] || !out[O_DUFLXC])) return fail(GEOSRAD_EINVAL, "dudTs set but duflx_dTs/duflxc_dTs null");
        bool any_bo = false;
        LwOut<R> O{};
        for (int b = 0; b < NB_LW; b++) { O.band_output[b] = band_output ? (band_output[b] != 0) : 0; any_bo |= O.band_output[b] != 0; }
        if (any_bo && (!out[O_OLRB] || (dudTs && !out[O_DOLRB]))) return fail(GEOSRAD_EINVAL, "band_output set but olrb/dolrb_dTs null");

        // one band's [layer][g<=16][column] plane of (a,bbu) pairs must stay below 4 GiB (32-bit byte offsets)
        const long cap = (long)(0xFFFFFFFFull / ((unsigned long long)nlay * 16ull * sizeof(R2))) & ~255L;
        int nc_max = ncol < chunk ? ncol : chunk;
        if ((long)nc_max > cap) nc_max = (int)cap;
        int rc = ensure_ws(nc_max, nlay);
        if (rc) return rc;
        R *rat_part = nullptr;
        if (rats && rats->n > 0) {
            if (rats->n > GEOSRAD_RAT_NGAS || !rats->uflx || !rats->dflx || (dudTs && !rats->duflx_dTs))
                return fail(GEOSRAD_EINVAL, "RATS: at most 8 gases; uflx_rat / dflx_rat (and duflx_dTs_rat with dudTs) must not be null");
            for (int r = 0; r < rats->n; r++)
                if (rats->gas[r] < 0 || rats->gas[r] >= GEOSRAD_RAT_NGAS) return fail(GEOSRAD_EINVAL, "RATS: unknown gas code");
            // an all-zero (nlay, ncol) plane stands for the removed gas's mixing ratio (and for pwvcm of a dry column); behind it
            // a second set of band partials, so that the main call's stay available to the bands a gas does not touch
            const size_t zplane = al((size_t)nlay * ncol * sizeof(R));
            const size_t need = zplane + (size_t)6 * NB_LW * (nlay + 1) * nc_max * sizeof(R);
            if (need > zero_bytes) {
                if (d_zero) { HIPCHK(hipFree(d_zero)); d_zero = nullptr; zero_bytes = 0; }
                if (hipMalloc((void **)&d_zero, need) != hipSuccess) return fail(GEOSRAD_ENOMEM, "hipMalloc of the RATS workspace failed");
                zero_bytes = need;
            }
            HIPCHK(hipMemsetAsync(d_zero, 0, zplane, st));
            rat_part = (R *)(d_zero + zplane);
        }

        for (int c0 = 0; c0 < ncol; c0 += nc_max) {
            const int nc = (ncol - c0) < nc_max ? (ncol - c0) : nc_max;
            Ws w;
            ws_layout(nc, nlay, &w, d_ws);
            LwArgs<R> A{};
            A.ncol = nc; A.ld = ncol; A.nlay = nlay; A.dudTs = dudTs; A.iceflg = iceflg; A.liqflg = liqflg; A.doy = dyofyr;
            A.cloudLM = cloudLM; A.cloudMH = cloudMH;
            auto P = [&](int k) { return in[k] ? (const R *)in[k] + c0 : (const R *)nullptr; };
            A.play = P(I_PLAY); A.plev = P(I_PLEV); A.tlay = P(I_TLAY); A.tlev = P(I_TLEV); A.tsfc = P(I_TSFC); A.emis = P(I_EMIS);
            A.h2o = P(I_H2O); A.o3 = P(I_O3); A.co2 = P(I_CO2); A.ch4 = P(I_CH4); A.n2o = P(I_N2O); A.o2 = P(I_O2);
            A.cfc11 = P(I_CFC11); A.cfc12 = P(I_CFC12); A.cfc22 = P(I_CFC22); A.ccl4 = P(I_CCL4);
            A.cldf = P(I_CLDF); A.ciwp = P(I_CIWP); A.clwp = P(I_CLWP); A.rei = P(I_REI); A.rel = P(I_REL);
            A.tauaer = P(I_TAUAER); A.zm = P(I_ZM); A.alat = P(I_ALAT);
            A.sc = w.sc; A.scidx = w.scidx; A.pwvcm = w.pwvcm; A.colcloudy = w.colcloudy; A.perm = w.perm; A.nclear = w.nclear; A.laycloudy = w.laycloudy;
            A.taucmc = w.taucmc; A.alpha = w.alpha; A.rcorr = w.rcorr; A.s1 = w.s1; A.s2 = w.s2; A.part = w.part;
            A.pfcode = w.pfcode; A.pffs = w.pffs;
            A.err = d_err;
            A.dbg_taug = dbg_taug ? (R *)dbg_taug + (size_t)c0 * NG_LW * nlay : nullptr;
            A.dbg_pfracs = dbg_pfracs ? (R *)dbg_pfracs + (size_t)c0 * NG_LW * nlay : nullptr;
            A.clearCounts = clearCounts + c0;
            A.band_mask = LW_ALL_BANDS;

            const dim3 blk(256);
            const unsigned gx = (unsigned)((nc + 255) / 256);
            span_begin(0, st); hipLaunchKernelGGL(k_validate_pwv<R>, dim3(gx), blk, 0, st, A, d_T);
            hipLaunchKernelGGL(k_partition, dim3(1), dim3(1024), 0, st, nc, (const uint8_t *)w.colcloudy, w.perm, w.nclear); span_end(st);
            span_begin(1, st); hipLaunchKernelGGL(k_setcoef<R>, dim3(gx, nlay), blk, 0, st, A, d_T); span_end(st);
            // McICA + cloud optics (threads of clear columns exit at once)
            span_begin(2, st); hipLaunchKernelGGL(k_overlap<R>, dim3(gx, nlay), blk, 0, st, nc, ncol, nlay, dyofyr, A.zm, A.alat,
                               (const int32_t *)w.perm, (const int32_t *)w.nclear, (const LwDev<R> *)d_T, A.alpha, A.rcorr, A.laycloudy); span_end(st);
            McArgs<R> M{};
            M.ncol = nc; M.ld = ncol; M.nlay = nlay; M.nsubcol = NG_LW; M.doy = dyofyr; M.cloudLM = cloudLM; M.cloudMH = cloudMH;
            M.iceflg = iceflg; M.liqflg = liqflg;
            M.so[0] = 1; M.so[1] = 2; M.so[2] = 3; M.so[3] = 4;        // seed_order=[1,2,3,4] (rrtmg_lw_rad.F90:546)
            M.cwp_tiny = (R)1.e-20;                                       // rrtmg_lw_rad.F90:544
            M.play = A.play; M.cldf = A.cldf; M.ciwp = A.ciwp; M.clwp = A.clwp; M.rei = A.rei; M.rel = A.rel;
            M.alpha = A.alpha; M.rcorr = A.rcorr; M.perm = w.perm; M.nclear = w.nclear; M.cftop = w.colcloudy;
            M.taucmc = A.taucmc; M.laycloudy = A.laycloudy; M.clearCounts = A.clearCounts; M.err = d_err;
            {
                McPlan MP; int nseg = 0;
                rc = mc_plan(0, NG_LW, nlay, MP, nseg);
                if (rc) return rc;
                span_begin(3, st);
                hipLaunchKernelGGL((k_mcica<R, 0>), dim3(xcd_grid(nc, 64, nseg)), dim3(64), 0, st, M, MP, (const LwDev<R> *)d_T, (const SwDev<R> *)nullptr);
                span_end(st);
            }
            auto Q = [&](int k) { return out[k] ? (R *)out[k] + c0 : (R *)nullptr; };
            O.uflx = Q(O_UFLX); O.dflx = Q(O_DFLX); O.uflxc = Q(O_UFLXC); O.dflxc = Q(O_DFLXC);
            O.duflx_dTs = Q(O_DUFLX); O.duflxc_dTs = Q(O_DUFLXC);
            O.olrb = (R *)out[O_OLRB]; O.dolrb_dTs = (R *)out[O_DOLRB]; O.col0 = c0;
            const size_t lds = lw_bands_lds_bytes<R>();
            // band sweeps.  lw_cols: (layer, g-point) intermediates in LDS, fluxes written directly (lw_cols_kernels.hpp);
            // lw_bands: lane = column with the parked cells (2-byte Pade indices) in HBM + the band reduction (the RATS passes need its per-band
            // partials, so a call with RATS diagnostics takes that path throughout)
            const bool cols = lw_cols_path && !(rats && rats->n > 0);
            span_begin(4, st);
            if (cols) {
                hipError_t e = lw_cols_launch<R>(st, A, O, h_T, A.dbg_taug != nullptr);
                if (e != hipSuccess) return fail(GEOSRAD_EHIP, std::string("lw_cols_launch: ") + hipGetErrorString(e));
            } else if (A.dbg_taug) {
                hipLaunchKernelGGL((k_lw_bands<R, true, true>), dim3(gx, NB_LW), blk, lds, st, A, h_T);
            } else if (lw_split_path) {
                hipError_t e = lw_split_launch<R>(st, A, h_T);
                if (e != hipSuccess) return fail(GEOSRAD_EHIP, std::string("lw_split_launch: ") + hipGetErrorString(e));
            } else {
                // (both instantiations band-major, heaviest band first: see band_block in lw_kernels.hpp)
                hipLaunchKernelGGL((k_lw_bands<R, false, false>), dim3(gx, NB_LW), blk, lds, st, A, h_T);
                if constexpr (sizeof(R) == 4)      // the 768-thread cloud-free blocks: run instead of the 256-thread ones when the batch has many cloud-free columns
                    hipLaunchKernelGGL((k_lw_bands<R, false, false, LW_WIDE_BLOCK>), dim3((unsigned)((nc + LW_WIDE_BLOCK - 1) / LW_WIDE_BLOCK), NB_LW),
                                       dim3(LW_WIDE_BLOCK), lds, st, A, h_T);
                hipLaunchKernelGGL((k_lw_bands<R, true, false>), dim3(gx, NB_LW), blk, lds, st, A, h_T);
            }
            span_end(st);
            if (!cols) { span_begin(5, st); hipLaunchKernelGGL(k_lw_reduce<R>, dim3(gx, nlay + 1), blk, 0, st, A, O); span_end(st); }

            // RATS diagnostics (GEOS_IrradGridComp.F90:3405-3468): the reference calls the whole of rrtmg_lw once more per listed
            // gas with that gas's mixing ratio set to zero and keeps the total-sky uflx, dflx, duflx_dTs of each call.  Nothing
            // the clouds decide depends on the gases: the input checks (zero passes them), the clear | cloudy partition, the
            // overlap correlations, the sub-columns with their cloud optical depths (`taucmc`, `laycloudy`) and clearCounts of
            // the batch are still in the workspace, so a gas costs setcoef + band sweeps + the reduction only - and only the bands
            // the gas appears in are swept again (LW_RAT_BANDS, lw_device.hpp): their partials go to a second buffer and the
            // reduction takes every other band's from the main call.  Without water vapour the precipitable water
            // (rrtmg_lw_setcoef.F90:206-272) is 0 / amttl = exactly zero.
            for (int r = 0; rats && r < rats->n; r++) {
                LwArgs<R> B = A;
                const R *z = (const R *)d_zero;
                switch (rats->gas[r]) {
                case GEOSRAD_RAT_H2O: B.h2o = z; B.pwvcm = (R *)d_zero; break;      // pwvcm is only read from here on
                case GEOSRAD_RAT_O3: B.o3 = z; break;
                case GEOSRAD_RAT_CO2: B.co2 = z; break;
                case GEOSRAD_RAT_CH4: B.ch4 = z; break;
                case GEOSRAD_RAT_N2O: B.n2o = z; break;
                case GEOSRAD_RAT_CFC11: B.cfc11 = z; break;
                case GEOSRAD_RAT_CFC12: B.cfc12 = z; break;
                default: B.cfc22 = z; break;
                }
                B.dbg_taug = nullptr; B.dbg_pfracs = nullptr;
                B.band_mask = LW_RAT_BANDS[rats->gas[r]];
                B.part = rat_part;
                span_begin(1, st); hipLaunchKernelGGL(k_setcoef<R>, dim3(gx, nlay), blk, 0, st, B, d_T); span_end(st);
                span_begin(4, st);
                if (lw_split_path) {
                    hipError_t e = lw_split_launch<R>(st, B, h_T);
                    if (e != hipSuccess) return fail(GEOSRAD_EHIP, std::string("lw_split_launch: ") + hipGetErrorString(e));
                } else {
                hipLaunchKernelGGL((k_lw_bands<R, false, false>), dim3(gx, NB_LW), blk, lds, st, B, h_T);
                if constexpr (sizeof(R) == 4)
                    hipLaunchKernelGGL((k_lw_bands<R, false, false, LW_WIDE_BLOCK>), dim3((unsigned)((nc + LW_WIDE_BLOCK - 1) / LW_WIDE_BLOCK), NB_LW),
                                       dim3(LW_WIDE_BLOCK), lds, st, B, h_T);
                hipLaunchKernelGGL((k_lw_bands<R, true, false>), dim3(gx, NB_LW), blk, lds, st, B, h_T);
                }
                span_end(st);
                LwOut<R> OR{};
                const size_t ro = (size_t)r * (nlay + 1) * ncol + c0;
                OR.uflx = (R *)rats->uflx + ro; OR.dflx = (R *)rats->dflx + ro;
                OR.duflx_dTs = rats->duflx_dTs ? (R *)rats->duflx_dTs + ro : nullptr;
                OR.col0 = c0; OR.part_alt = rat_part; OR.alt_mask = B.band_mask;
                B.part = A.part;
                span_begin(5, st); hipLaunchKernelGGL(k_lw_reduce<R>, dim3(gx, nlay + 1), blk, 0, st, B, OR); span_end(st);
            }
        }
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    // ---- GridComp drivers (gridcomp_kernels.hpp) ---------------------------------------------------------------------------
    int drv_reserve(int which, size_t need)
    {
        if (need <= ws_drvs_bytes[which]) return GEOSRAD_OK;
        if (d_ws_drvs[which]) { HIPCHK(hipFree(d_ws_drvs[which])); d_ws_drvs[which] = nullptr; ws_drvs_bytes[which] = 0; }
        if (hipMalloc((void **)&d_ws_drvs[which], need) != hipSuccess) return fail(GEOSRAD_ENOMEM, "hipMalloc of the driver workspace failed");
        ws_drvs_bytes[which] = need;
        return GEOSRAD_OK;
    }

    int lw_driver_dev(hipStream_t st, int ncol, int lm, int nb, const void *const *in, const double *consts, int iceflg, int liqflg,
                      int doy, int lcldlm, int lcldmh, const int32_t *band_output, void *const *out, int nrats, const int32_t *rat_gas,
                      void *const *rat_out) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || lm < 4 || nb < 0 || nb > 16) return fail(GEOSRAD_EINVAL, "bad ncol/lm/nb_aer");
        if (nrats < 0 || nrats > GEOSRAD_RAT_NGAS || (nrats > 0 && (!rat_gas || !rat_out))) return fail(GEOSRAD_EINVAL, "bad RATS arguments");
        for (int k = 0; k < GEOSRAD_LWD_NIN; k++)
            if (!in[k] && k != GEOSRAD_LWD_CO2_3D && k != GEOSRAD_LWD_TAUA && k != GEOSRAD_LWD_SSAA) return fail(GEOSRAD_EINVAL, "null input array");
        if ((in[GEOSRAD_LWD_TAUA] == nullptr) != (in[GEOSRAD_LWD_SSAA] == nullptr)) return fail(GEOSRAD_EINVAL, "TAUA and SSAA go together");
        const size_t n = (size_t)ncol, cl = n * lm, cv = n * (lm + 1);
        size_t off = 0;
        auto take = [&](size_t nreal) { size_t o = off; off += al(nreal * sizeof(R)); return o; };
        size_t o_lay[18], o_lev[2], o_flux[6];
        for (auto &o : o_lay) o = take(cl);
        for (auto &o : o_lev) o = take(cv);
        const size_t o_tsfc = take(n), o_alat = take(n), o_emis = take(n * 16), o_aer = take(cl * 16);
        for (auto &o : o_flux) o = take(cv);
        const size_t o_olrb = take(n * 16), o_dolrb = take(n * 16), o_cc = take(n * 4);
        size_t o_rat[3] = {0, 0, 0};
        for (auto &o : o_rat) o = take(cv * (size_t)nrats);
        int rc = drv_reserve(0, off);
        if (rc) return rc;
        char *const d_ws_drv = d_ws_drvs[0];
        auto P = [&](size_t o) { return (R *)(d_ws_drv + o); };
        LwdArgs<R> A{};
        A.ncol = ncol; A.lm = lm; A.nb = in[GEOSRAD_LWD_TAUA] ? nb : 0; A.iceflg = iceflg; A.liqflg = liqflg;
        auto I = [&](int k) { return (const R *)in[k]; };
        A.ple = I(GEOSRAD_LWD_PLE); A.pl = I(GEOSRAD_LWD_PL); A.t = I(GEOSRAD_LWD_T); A.q = I(GEOSRAD_LWD_Q); A.o3 = I(GEOSRAD_LWD_O3);
        A.ch4 = I(GEOSRAD_LWD_CH4); A.n2o = I(GEOSRAD_LWD_N2O); A.co2_3d = I(GEOSRAD_LWD_CO2_3D); A.cfc11 = I(GEOSRAD_LWD_CFC11);
        A.cfc12 = I(GEOSRAD_LWD_CFC12); A.hcfc22 = I(GEOSRAD_LWD_HCFC22); A.fcld = I(GEOSRAD_LWD_FCLD);
        A.cwc_liq = I(GEOSRAD_LWD_CWC_LIQ); A.cwc_ice = I(GEOSRAD_LWD_CWC_ICE); A.reff_liq = I(GEOSRAD_LWD_REFF_LIQ);
        A.reff_ice = I(GEOSRAD_LWD_REFF_ICE); A.taua = I(GEOSRAD_LWD_TAUA); A.ssaa = I(GEOSRAD_LWD_SSAA); A.ts = I(GEOSRAD_LWD_TS);
        A.emis = I(GEOSRAD_LWD_EMIS); A.lats = I(GEOSRAD_LWD_LATS); A.t2m = I(GEOSRAD_LWD_T2M);
        A.co2_fixed = (R)consts[GEOSRAD_LWD_C_CO2_FIXED]; A.o2 = (R)consts[GEOSRAD_LWD_C_O2]; A.ccl4 = (R)consts[GEOSRAD_LWD_C_CCL4];
        // (MAPL_AIRMW/MAPL_H2OMW), (MAPL_AIRMW/MAPL_O3MW): constant expressions of the caller's real kind
        A.airmw_over_h2omw = (R)consts[GEOSRAD_C_AIRMW] / (R)consts[GEOSRAD_C_H2OMW];
        A.airmw_over_o3mw = (R)consts[GEOSRAD_C_AIRMW] / (R)consts[GEOSRAD_C_O3MW];
        A.rgas = (R)consts[GEOSRAD_C_RGAS]; A.grav = (R)consts[GEOSRAD_C_GRAV];
        A.play = P(o_lay[0]); A.tlay = P(o_lay[1]); A.h2o = P(o_lay[2]); A.o3_r = P(o_lay[3]); A.co2_r = P(o_lay[4]); A.ch4_r = P(o_lay[5]);
        A.n2o_r = P(o_lay[6]); A.o2_r = P(o_lay[7]); A.cfc11_r = P(o_lay[8]); A.cfc12_r = P(o_lay[9]); A.cfc22_r = P(o_lay[10]);
        A.ccl4_r = P(o_lay[11]); A.cldf = P(o_lay[12]); A.ciwp = P(o_lay[13]); A.clwp = P(o_lay[14]); A.rei = P(o_lay[15]);
        A.rel = P(o_lay[16]); A.zm = P(o_lay[17]); A.plev = P(o_lev[0]); A.tlev = P(o_lev[1]); A.tsfc = P(o_tsfc); A.alat = P(o_alat);
        A.emis_r = P(o_emis); A.tauaer = P(o_aer);
        const dim3 blk(256);
        const unsigned gx = (unsigned)((ncol + 255) / 256);
        hipLaunchKernelGGL((k_lwd_prep<R>), dim3(gx, lm), blk, 0, st, A);
        hipLaunchKernelGGL((k_lwd_zm<R>), dim3(gx), blk, 0, st, A);
        // reverse the super-layer interface indices (IRR:3237-3239) and call the solver with Ts_derivs = .true.
        const int cloudMH = lm - lcldmh + 1, cloudLM = lm - lcldlm + 1;
        const void *lin[I_NIN];
        lin[I_PLAY] = A.play; lin[I_PLEV] = A.plev; lin[I_TLAY] = A.tlay; lin[I_TLEV] = A.tlev; lin[I_TSFC] = A.tsfc; lin[I_EMIS] = A.emis_r;
        lin[I_H2O] = A.h2o; lin[I_O3] = A.o3_r; lin[I_CO2] = A.co2_r; lin[I_CH4] = A.ch4_r; lin[I_N2O] = A.n2o_r; lin[I_O2] = A.o2_r;
        lin[I_CFC11] = A.cfc11_r; lin[I_CFC12] = A.cfc12_r; lin[I_CFC22] = A.cfc22_r; lin[I_CCL4] = A.ccl4_r; lin[I_CLDF] = A.cldf;
        lin[I_CIWP] = A.ciwp; lin[I_CLWP] = A.clwp; lin[I_REI] = A.rei; lin[I_REL] = A.rel; lin[I_TAUAER] = A.tauaer; lin[I_ZM] = A.zm;
        lin[I_ALAT] = A.alat;
        void *lout[O_NOUT] = {P(o_flux[0]), P(o_flux[1]), P(o_flux[2]), P(o_flux[3]), P(o_flux[4]), P(o_flux[5]),
                              out[GEOSRAD_LWD_OLRB] ? out[GEOSRAD_LWD_OLRB] : (void *)P(o_olrb),
                              out[GEOSRAD_LWD_DOLRB] ? out[GEOSRAD_LWD_DOLRB] : (void *)P(o_dolrb)};
        int32_t *cc = (int32_t *)(d_ws_drv + o_cc);
        static const int32_t no_bands[16] = {0};
        LwRats RT{};
        RT.n = nrats; RT.uflx = P(o_rat[0]); RT.dflx = P(o_rat[1]); RT.duflx_dTs = P(o_rat[2]);
        for (int r = 0; r < nrats; r++) RT.gas[r] = rat_gas[r];
        rc = lw_dev(st, ncol, lm, 1, lin, iceflg, liqflg, doy, cloudLM, cloudMH, cc, lout, band_output ? band_output : no_bands, nullptr,
                    nullptr, nrats > 0 ? &RT : nullptr);
        if (rc) return rc;
        if (nrats > 0) {
            LwdRatPost<R> RP{};
            RP.ncol = ncol; RP.lm = lm; RP.nrats = nrats; RP.uflx = P(o_rat[0]); RP.dflx = P(o_rat[1]); RP.duflx = P(o_rat[2]); RP.emis = A.emis;
            RP.flxu_rat = (R *)rat_out[GEOSRAD_LWD_FLXU_RAT]; RP.flxd_rat = (R *)rat_out[GEOSRAD_LWD_FLXD_RAT];
            RP.flx_rat = (R *)rat_out[GEOSRAD_LWD_FLX_RAT]; RP.dfdts_rat = (R *)rat_out[GEOSRAD_LWD_DFDTS_RAT];
            RP.sfcem_rat = (R *)rat_out[GEOSRAD_LWD_SFCEM_RAT];
            hipLaunchKernelGGL((k_lwd_rat_post<R>), dim3(gx, lm + 1, nrats), blk, 0, st, RP);
        }
        LwdPost<R> Q{};
        Q.ncol = ncol; Q.lm = lm; Q.ngpt = NG_LW;
        Q.uflx = P(o_flux[0]); Q.dflx = P(o_flux[1]); Q.uflxc = P(o_flux[2]); Q.dflxc = P(o_flux[3]); Q.duflx = P(o_flux[4]);
        Q.duflxc = P(o_flux[5]); Q.clearCounts = cc; Q.emis = A.emis; Q.ts = A.ts;
        auto O = [&](int k) { return (R *)out[k]; };
        Q.flxu_int = O(GEOSRAD_LWD_FLXU_INT); Q.flxd_int = O(GEOSRAD_LWD_FLXD_INT); Q.flcu_int = O(GEOSRAD_LWD_FLCU_INT);
        Q.flcd_int = O(GEOSRAD_LWD_FLCD_INT); Q.dfdts = O(GEOSRAD_LWD_DFDTS); Q.dfdtsc = O(GEOSRAD_LWD_DFDTSC);
        Q.dfdtsna = O(GEOSRAD_LWD_DFDTSNA); Q.dfdtscna = O(GEOSRAD_LWD_DFDTSCNA); Q.flx_int = O(GEOSRAD_LWD_FLX_INT);
        Q.flc_int = O(GEOSRAD_LWD_FLC_INT); Q.sfcem_int = O(GEOSRAD_LWD_SFCEM_INT); Q.ts_int = O(GEOSRAD_LWD_TS_INT);
        Q.cldttlw = O(GEOSRAD_LWD_CLDTTLW); Q.cldhilw = O(GEOSRAD_LWD_CLDHILW); Q.cldmdlw = O(GEOSRAD_LWD_CLDMDLW);
        Q.cldlolw = O(GEOSRAD_LWD_CLDLOLW);
        hipLaunchKernelGGL((k_lwd_post<R>), dim3(gx, lm + 1), blk, 0, st, Q);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int sw_driver_dev(hipStream_t st, int ncol, int lm, int nb, const void *const *in, const double *consts, int iceflg, int liqflg,
                      double sc, double dist, int isolvar, int dyofyr, int include_aerosols, int lcldlm, int lcldmh, int normflx,
                      const void *bndsolvar, const void *indsolvar, void *const *out) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || lm < 4 || nb < 0 || nb > 14) return fail(GEOSRAD_EINVAL, "bad ncol/lm/nb_aer");
        // SORADCORE asserts the solar-variability options it supports before the call (GEOS_SolarGridComp.F90:6286-6292): no isolvar 1
        if (isolvar == 1) return fail(GEOSRAD_EINPUT, "SORADCORE: ISOLVAR == 1 is not supported by the GridComp (the solver entry point rrtmg_sw accepts it)");
        for (int k = 0; k < GEOSRAD_SWD_NIN; k++)
            if (!in[k] && k != GEOSRAD_SWD_TAUA && k != GEOSRAD_SWD_SSAA && k != GEOSRAD_SWD_ASYA) return fail(GEOSRAD_EINVAL, "null input array");
        const bool aer = in[GEOSRAD_SWD_TAUA] != nullptr;
        if (aer && (!in[GEOSRAD_SWD_SSAA] || !in[GEOSRAD_SWD_ASYA])) return fail(GEOSRAD_EINVAL, "TAUA, SSAA and ASYA go together");
        if (aer && nb != 14) return fail(GEOSRAD_EINVAL, "RRTMG_SW aerosol arrays have 14 bands");
        const size_t n = (size_t)ncol, cl = n * lm, cv = n * (lm + 1);
        size_t off = 0;
        auto take = [&](size_t nreal) { size_t o = off; off += al(nreal * sizeof(R)); return o; };
        const bool want_na = out[GEOSRAD_SWD_FSWNA] || out[GEOSRAD_SWD_FSCNA] || out[GEOSRAD_SWD_FSWUNA] || out[GEOSRAD_SWD_FSCUNA] ||
                             out[GEOSRAD_SWD_FSWBANDNA];
        size_t o_lay[13], o_lev[2], o_aer[3], o_flux[4], o_sc[6], o_cot[8], o_nflux[4], o_nsc[14];
        for (auto &o : o_lay) o = take(cl);
        for (auto &o : o_lev) o = take(cv);
        for (auto &o : o_aer) o = take(cl * 14);
        for (auto &o : o_flux) o = take(cv);
        for (auto &o : o_sc) o = take(n);
        for (auto &o : o_cot) o = take(n);
        const size_t o_band = take(n * 14), o_cc = take(n * 4);
        for (auto &o : o_nflux) o = want_na ? take(cv) : 0;
        for (auto &o : o_nsc) o = want_na ? take(n) : 0;
        const size_t o_nband = want_na ? take(n * 14) : 0;
        int rc = drv_reserve(1, off);
        if (rc) return rc;
        char *const d_ws_drv = d_ws_drvs[1];
        auto P = [&](size_t o) { return (R *)(d_ws_drv + o); };
        auto I = [&](int k) { return (const R *)in[k]; };
        SwdArgs<R> A{};
        A.ncol = ncol; A.lm = lm; A.nb = 14; A.iceflg = iceflg; A.liqflg = liqflg;
        A.ple = I(GEOSRAD_SWD_PLE); A.pl = I(GEOSRAD_SWD_PL); A.t = I(GEOSRAD_SWD_T); A.q = I(GEOSRAD_SWD_Q); A.o3 = I(GEOSRAD_SWD_O3);
        A.ch4 = I(GEOSRAD_SWD_CH4); A.cl = I(GEOSRAD_SWD_CL); A.ts = I(GEOSRAD_SWD_TS); A.qq_ice = I(GEOSRAD_SWD_QQ_ICE);
        A.qq_liq = I(GEOSRAD_SWD_QQ_LIQ); A.rr_ice = I(GEOSRAD_SWD_RR_ICE); A.rr_liq = I(GEOSRAD_SWD_RR_LIQ);
        A.taua = (R *)in[GEOSRAD_SWD_TAUA]; A.ssaa = (R *)in[GEOSRAD_SWD_SSAA]; A.asya = (R *)in[GEOSRAD_SWD_ASYA];
        A.co2 = (R)consts[GEOSRAD_SWD_C_CO2]; A.o2 = (R)consts[GEOSRAD_SWD_C_O2];
        A.airmw_over_h2omw = (R)consts[GEOSRAD_SWD_C_AIRMW] / (R)consts[GEOSRAD_SWD_C_H2OMW];
        A.airmw_over_o3mw = (R)consts[GEOSRAD_SWD_C_AIRMW] / (R)consts[GEOSRAD_SWD_C_O3MW];
        A.rgas = (R)consts[GEOSRAD_SWD_C_RGAS]; A.grav = (R)consts[GEOSRAD_SWD_C_GRAV];
        A.play = P(o_lay[0]); A.tlay = P(o_lay[1]); A.h2o = P(o_lay[2]); A.o3_r = P(o_lay[3]); A.co2_r = P(o_lay[4]); A.ch4_r = P(o_lay[5]);
        A.o2_r = P(o_lay[6]); A.cldf = P(o_lay[7]); A.ciwp = P(o_lay[8]); A.clwp = P(o_lay[9]); A.rei = P(o_lay[10]); A.rel = P(o_lay[11]);
        A.zl = P(o_lay[12]); A.plev = P(o_lev[0]); A.tlev = P(o_lev[1]); A.tauaer = P(o_aer[0]); A.ssaaer = P(o_aer[1]); A.asmaer = P(o_aer[2]);
        const dim3 blk(256);
        const unsigned gx = (unsigned)((ncol + 255) / 256);
        hipLaunchKernelGGL((k_swd_prep<R>), dim3(gx, lm), blk, 0, st, A);
        hipLaunchKernelGGL((k_swd_zm<R>), dim3(gx), blk, 0, st, A);
        const void *sin[S_NIN];
        sin[S_PLAY] = A.play; sin[S_PLEV] = A.plev; sin[S_TLAY] = A.tlay; sin[S_H2O] = A.h2o; sin[S_O3] = A.o3_r; sin[S_CO2] = A.co2_r;
        sin[S_CH4] = A.ch4_r; sin[S_O2] = A.o2_r; sin[S_CLD] = A.cldf; sin[S_CIWP] = A.ciwp; sin[S_CLWP] = A.clwp; sin[S_REI] = A.rei;
        sin[S_REL] = A.rel; sin[S_ZM] = A.zl; sin[S_ALAT] = in[GEOSRAD_SWD_ALAT]; sin[S_TAUAER] = A.tauaer; sin[S_SSAAER] = A.ssaaer;
        sin[S_ASMAER] = A.asmaer; sin[S_COSZEN] = in[GEOSRAD_SWD_ZT]; sin[S_ASDIR] = in[GEOSRAD_SWD_ALBVR]; sin[S_ASDIF] = in[GEOSRAD_SWD_ALBVF];
        sin[S_ALDIR] = in[GEOSRAD_SWD_ALBNR]; sin[S_ALDIF] = in[GEOSRAD_SWD_ALBNF];
        void *sout[SO_NOUT] = {};
        for (int k = 0; k < 4; k++) sout[SO_UFLX + k] = P(o_flux[k]);
        const int sc_ix[6] = {GEOSRAD_SWD_NIRR, GEOSRAD_SWD_NIRF, GEOSRAD_SWD_PARR, GEOSRAD_SWD_PARF, GEOSRAD_SWD_UVRR, GEOSRAD_SWD_UVRF};
        for (int k = 0; k < 6; k++) sout[SO_NIRR + k] = out[sc_ix[k]] ? out[sc_ix[k]] : (void *)P(o_sc[k]);
        sout[SO_FSWBAND] = out[GEOSRAD_SWD_FSWBAND] ? out[GEOSRAD_SWD_FSWBAND] : (void *)P(o_band);
        for (int k = 0; k < 8; k++) sout[SO_COT0 + k] = P(o_cot[k]);      // cotd t/h/m/l then cotn t/h/m/l
        int32_t *cc = (int32_t *)(d_ws_drv + o_cc);
        // IAER = 10 always (SOL:6235; without aerosols the arrays are zero); super-layer indices flipped in the call (SOL:6341)
        void *nout[SO_NOUT] = {};
        if (want_na) {
            for (int k = 0; k < 4; k++) nout[SO_UFLX + k] = P(o_nflux[k]);
            for (int k = 0; k < 6; k++) nout[SO_NIRR + k] = P(o_nsc[k]);
            for (int k = 0; k < 8; k++) nout[SO_COT0 + k] = P(o_nsc[6 + k]);
            nout[SO_FSWBAND] = out[GEOSRAD_SWD_FSWBANDNA] ? out[GEOSRAD_SWD_FSWBANDNA] : (void *)P(o_nband);
        }
        rc = sw_run(st, ncol, lm, sc, dist, isolvar, sin, iceflg, liqflg, dyofyr, 10, lm - lcldlm + 1, lm - lcldmh + 1,
                    normflx, cc, sout, 0, bndsolvar, indsolvar, nullptr, nullptr, want_na ? nout : nullptr);
        if (rc) return rc;
        SwdPost<R> Q{};
        Q.ncol = ncol; Q.lm = lm; Q.ngpt = NG_SW; Q.aerosols = include_aerosols; Q.undef = (R)consts[GEOSRAD_SWD_C_UNDEF];
        Q.swuflx = P(o_flux[0]); Q.swdflx = P(o_flux[1]); Q.swuflxc = P(o_flux[2]); Q.swdflxc = P(o_flux[3]); Q.clearCounts = cc;
        for (int k = 0; k < 4; k++) { Q.cotd[k] = P(o_cot[k]); Q.cotn[k] = P(o_cot[4 + k]); Q.cot[k] = (R *)out[GEOSRAD_SWD_COTTP + k]; }
        Q.fsw = (R *)out[GEOSRAD_SWD_FSW]; Q.fsc = (R *)out[GEOSRAD_SWD_FSC]; Q.fswu = (R *)out[GEOSRAD_SWD_FSWU]; Q.fscu = (R *)out[GEOSRAD_SWD_FSCU];
        Q.cldts = (R *)out[GEOSRAD_SWD_CLDTS]; Q.cldhs = (R *)out[GEOSRAD_SWD_CLDHS]; Q.cldms = (R *)out[GEOSRAD_SWD_CLDMS];
        Q.cldls = (R *)out[GEOSRAD_SWD_CLDLS];
        hipLaunchKernelGGL((k_swd_post<R>), dim3(gx, lm + 1), blk, 0, st, Q);
        if (want_na) {      // un-flip of the no-aerosol fluxes (the FS*NAN internals, SOL:4152-4159)
            SwdPost<R> N{};
            N.ncol = ncol; N.lm = lm; N.ngpt = NG_SW; N.aerosols = 0; N.undef = Q.undef;
            N.swuflx = P(o_nflux[0]); N.swdflx = P(o_nflux[1]); N.swuflxc = P(o_nflux[2]); N.swdflxc = P(o_nflux[3]); N.clearCounts = cc;
            N.fsw = (R *)out[GEOSRAD_SWD_FSWNA]; N.fsc = (R *)out[GEOSRAD_SWD_FSCNA]; N.fswu = (R *)out[GEOSRAD_SWD_FSWUNA];
            N.fscu = (R *)out[GEOSRAD_SWD_FSCUNA];
            hipLaunchKernelGGL((k_swd_post<R>), dim3(gx, lm + 1), blk, 0, st, N);
        }
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int lw_chou_post_dev(hipStream_t st, int ncol, int lm, const void *const *in, void *const *out) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || lm <= 0) return fail(GEOSRAD_EINVAL, "bad ncol/lm");
        LwcPost<R> P{};
        P.ncol = ncol; P.lm = lm;
        static_assert(offsetof(LwcPost<R>, ts) - offsetof(LwcPost<R>, flxu) == (GEOSRAD_LWC_NIN - 1) * sizeof(void *), "LwcPost input layout");
        static_assert(offsetof(LwcPost<R>, ts_int) - offsetof(LwcPost<R>, sfcem_int) == (GEOSRAD_LWC_NOUT - 1) * sizeof(void *), "LwcPost output layout");
        const R **ip = &P.flxu;
        for (int k = 0; k < GEOSRAD_LWC_NIN; k++) ip[k] = (const R *)in[k];
        R **op = &P.sfcem_int;
        for (int k = 0; k < GEOSRAD_LWC_NOUT; k++) op[k] = (R *)out[k];
        auto need = [&](const R *o, const R *a, const R *b = (const R *)1) { return !o || (a && b); };
        if (!(need(P.flx_int, P.flxd, P.flxu) && need(P.flxa_int, P.flxad, P.flxau) && need(P.flc_int, P.flcd, P.flcu) &&
              need(P.fla_int, P.flad, P.flau) && need(P.dfdtsna, P.dfdts) && need(P.ts_int, P.ts)))
            return fail(GEOSRAD_EINVAL, "an output was requested without the field it is computed from");
        hipLaunchKernelGGL((k_lwd_chou_post<R>), dim3((unsigned)((ncol + 255) / 256), lm + 1), dim3(256), 0, st, P);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    // Chou-Suarez branch of SORADCORE: k_swc_prep + sorad_dev.  The prepared arrays live in a buffer of their own (sorad_dev's scratch is
    // sized per chunk, these per call).
    int sw_driver_chou_dev(hipStream_t st, int ncol, int lm, const void *const *in, const double *consts, int lcldmh, int lcldlm,
                           const void *hk_uv, const void *hk_ir, int do_drfband, void *const *out) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || lm < 4) return fail(GEOSRAD_EINVAL, "bad ncol/lm");
        if (!consts) return fail(GEOSRAD_EINVAL, "consts null");
        const bool aer = in[GEOSRAD_SWC_TAUA] != nullptr;
        if (aer != (in[GEOSRAD_SWC_SSAA] != nullptr) || aer != (in[GEOSRAD_SWC_ASYA] != nullptr))
            return fail(GEOSRAD_EINVAL, "TAUA / SSAA / ASYA: all three or none");
        for (int k = 0; k < GEOSRAD_SWC_NIN; k++)
            if (!in[k] && !(k >= GEOSRAD_SWC_TAUA && k <= GEOSRAD_SWC_ASYA)) return fail(GEOSRAD_EINVAL, "null input field");
        const size_t cell = (size_t)ncol * sizeof(R);
        const size_t o_plh = 0, o_o3 = o_plh + al((size_t)(lm + 1) * cell), o_qq = o_o3 + al((size_t)lm * cell), o_rr = o_qq + al((size_t)4 * lm * cell),
                     o_zero = o_rr + al((size_t)4 * lm * cell), need = o_zero + (aer ? 0 : al((size_t)8 * lm * cell));
        if (need > ws_swc_bytes) {
            if (d_ws_swc) { HIPCHK(hipFree(d_ws_swc)); d_ws_swc = nullptr; ws_swc_bytes = 0; }
            if (hipMalloc((void **)&d_ws_swc, need) != hipSuccess) return fail(GEOSRAD_ENOMEM, "SORADCORE (Chou-Suarez) workspace");
            ws_swc_bytes = need;
        }
        SwcPrep<R> P{};
        P.ncol = ncol; P.lm = lm;
        P.ple = (const R *)in[GEOSRAD_SWC_PLE]; P.ox = (const R *)in[GEOSRAD_SWC_OX];
        for (int s = 0; s < 4; s++) { P.q[s] = (const R *)in[GEOSRAD_SWC_QI + s]; P.r[s] = (const R *)in[GEOSRAD_SWC_RI + s]; }
        P.o3fac = (R)consts[GEOSRAD_SWC_C_O3MW] / (R)consts[GEOSRAD_SWC_C_AIRMW]; P.undef = (R)consts[GEOSRAD_SWC_C_UNDEF];
        P.plhpa = (R *)(d_ws_swc + o_plh); P.o3 = (R *)(d_ws_swc + o_o3); P.qq3 = (R *)(d_ws_swc + o_qq); P.rr3 = (R *)(d_ws_swc + o_rr);
        hipLaunchKernelGGL((k_swc_prep<R>), dim3((unsigned)((ncol + 255) / 256), lm + 1), dim3(256), 0, st, P);
        HIPCHK(hipGetLastError());
        const void *zero = d_ws_swc + o_zero;          // TAUA = SSAA = ASYA = 0 (SOL:4543-4546): one block serves the three
        if (!aer) HIPCHK(hipMemsetAsync(d_ws_swc + o_zero, 0, (size_t)8 * lm * cell, st));
        const void *si[SI_NIN];
        si[SI_COSZ] = in[GEOSRAD_SWC_ZT]; si[SI_PL] = P.plhpa; si[SI_TA] = in[GEOSRAD_SWC_T]; si[SI_WA] = in[GEOSRAD_SWC_Q]; si[SI_OA] = P.o3;
        si[SI_CWC] = P.qq3; si[SI_FCLD] = in[GEOSRAD_SWC_CL]; si[SI_REFF] = P.rr3;
        si[SI_TAUA] = aer ? in[GEOSRAD_SWC_TAUA] : zero; si[SI_SSAA] = aer ? in[GEOSRAD_SWC_SSAA] : zero; si[SI_ASYA] = aer ? in[GEOSRAD_SWC_ASYA] : zero;
        si[SI_RSUVBM] = in[GEOSRAD_SWC_ALBVR]; si[SI_RSUVDF] = in[GEOSRAD_SWC_ALBVF]; si[SI_RSIRBM] = in[GEOSRAD_SWC_ALBNR]; si[SI_RSIRDF] = in[GEOSRAD_SWC_ALBNF];
        void *so[SOO_NOUT];
        so[SOO_FLX] = out[GEOSRAD_SWC_FSW]; so[SOO_FLC] = out[GEOSRAD_SWC_FSC]; so[SOO_FLXU] = out[GEOSRAD_SWC_FSWU]; so[SOO_FLCU] = out[GEOSRAD_SWC_FSCU];
        so[SOO_FDIRIR] = out[GEOSRAD_SWC_NIRR]; so[SOO_FDIFIR] = out[GEOSRAD_SWC_NIRF]; so[SOO_FDIRPAR] = out[GEOSRAD_SWC_PARR];
        so[SOO_FDIFPAR] = out[GEOSRAD_SWC_PARF]; so[SOO_FDIRUV] = out[GEOSRAD_SWC_UVRR]; so[SOO_FDIFUV] = out[GEOSRAD_SWC_UVRF];
        so[SOO_SFCBAND] = out[GEOSRAD_SWC_FSWBAND]; so[SOO_DRBAND] = out[GEOSRAD_SWC_DRBAND]; so[SOO_DFBAND] = out[GEOSRAD_SWC_DFBAND];
        return sorad_dev(st, ncol, lm, 8, si, consts[GEOSRAD_SWC_C_CO2], lcldmh, lcldlm, hk_uv, hk_ir, so, do_drfband);
    }

    int lw_update_flx_dev(hipStream_t st, int ncol, int lm, int rrtmg, int lev_mid_high, int lev_low_mid, double undef,
                          const void *const *in, void *const *out) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || lm <= 0) return fail(GEOSRAD_EINVAL, "bad ncol/lm");
        if (lev_mid_high < 1 || lev_mid_high > lm || lev_low_mid < 1 || lev_low_mid > lm) return fail(GEOSRAD_EINVAL, "bad super-layer levels");
        auto na = [](int k) {
            return k == GEOSRAD_LWU_FLXA_INT || k == GEOSRAD_LWU_FLA_INT || k == GEOSRAD_LWU_FLXAU_INT || k == GEOSRAD_LWU_FLAU_INT ||
                   k == GEOSRAD_LWU_FLXAD_INT || k == GEOSRAD_LWU_FLAD_INT || k == GEOSRAD_LWU_DFDTSNA || k == GEOSRAD_LWU_DFDTSCNA;
        };
        for (int k = 0; k < GEOSRAD_LWU_NIN; k++)
            if (!in[k] && !(rrtmg && na(k))) return fail(GEOSRAD_EINVAL, "null internal-state array");
        LwUpd<R> U{};
        U.ncol = ncol; U.lm = lm; U.rrtmg = rrtmg; U.lev_mid_high = lev_mid_high; U.lev_low_mid = lev_low_mid; U.undef = (R)undef;
        auto I = [&](int k) { return (const R *)in[k]; };
        U.tsinst = I(GEOSRAD_LWU_TSINST); U.ts_int = I(GEOSRAD_LWU_TS_INT); U.sfcem_int = I(GEOSRAD_LWU_SFCEM_INT); U.fcld = I(GEOSRAD_LWU_FCLD);
        U.flx_int = I(GEOSRAD_LWU_FLX_INT); U.flxa_int = I(GEOSRAD_LWU_FLXA_INT); U.flc_int = I(GEOSRAD_LWU_FLC_INT); U.fla_int = I(GEOSRAD_LWU_FLA_INT);
        U.flxu_int = I(GEOSRAD_LWU_FLXU_INT); U.flxau_int = I(GEOSRAD_LWU_FLXAU_INT); U.flcu_int = I(GEOSRAD_LWU_FLCU_INT);
        U.flau_int = I(GEOSRAD_LWU_FLAU_INT); U.flxd_int = I(GEOSRAD_LWU_FLXD_INT); U.flxad_int = I(GEOSRAD_LWU_FLXAD_INT);
        U.flcd_int = I(GEOSRAD_LWU_FLCD_INT); U.flad_int = I(GEOSRAD_LWU_FLAD_INT); U.dfdts = I(GEOSRAD_LWU_DFDTS);
        U.dfdtsna = I(GEOSRAD_LWU_DFDTSNA); U.dfdtsc = I(GEOSRAD_LWU_DFDTSC); U.dfdtscna = I(GEOSRAD_LWU_DFDTSCNA);
        static_assert(offsetof(LwUpd<R>, cldtt) - offsetof(LwUpd<R>, flx) == (GEOSRAD_LWU_NOUT - 1) * sizeof(void *), "LwUpd export layout");
        R **o3 = &U.flx;       // the export pointers are laid out in the order of the GEOSRAD_LWU_* output enum
        for (int k = 0; k < GEOSRAD_LWU_NOUT; k++) o3[k] = (R *)out[k];
        // 16-byte accesses (4 floats / 2 doubles per thread) when the column count and every field address allow
        constexpr int VW = 16 / (int)sizeof(R);
        bool wide = ncol % VW == 0;
        for (int k = 0; k < GEOSRAD_LWU_NIN; k++) wide = wide && ((uintptr_t)in[k] & 15) == 0;
        for (int k = 0; k < GEOSRAD_LWU_NOUT; k++) wide = wide && ((uintptr_t)out[k] & 15) == 0;
        if (wide) hipLaunchKernelGGL((k_lw_update_flx<R, VW>), dim3((unsigned)((ncol / VW + 255) / 256), lm + 1), dim3(256), 0, st, U);
        else hipLaunchKernelGGL((k_lw_update_flx<R, 1>), dim3((unsigned)((ncol + 255) / 256), lm + 1), dim3(256), 0, st, U);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int lw_update_rats_dev(hipStream_t st, int ncol, int lm, int nrats, const void *const *in, void *const *out) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || lm <= 0 || nrats < 0 || nrats > GEOSRAD_RAT_NGAS) return fail(GEOSRAD_EINVAL, "bad ncol/lm/nrats");
        if (nrats == 0) return GEOSRAD_OK;
        for (int k = 0; k < GEOSRAD_LWR_NIN; k++)
            if (!in[k] && !(k == GEOSRAD_LWR_DFDTS || k == GEOSRAD_LWR_DFDTS_RAT)) return fail(GEOSRAD_EINVAL, "null internal-state array");
        if (out[GEOSRAD_LWR_DFDTS_OUT] && (!in[GEOSRAD_LWR_DFDTS] || !in[GEOSRAD_LWR_DFDTS_RAT]))
            return fail(GEOSRAD_EINVAL, "DFDTS_<gas> requested without DFDTS / DFDTS_RAT");
        LwRatUpd<R> U{};
        U.ncol = ncol; U.lm = lm; U.nrats = nrats;
        auto I = [&](int k) { return (const R *)in[k]; };
        U.flx_int = I(GEOSRAD_LWR_FLX_INT); U.sfcem_int = I(GEOSRAD_LWR_SFCEM_INT); U.dfdts = I(GEOSRAD_LWR_DFDTS);
        U.flx_rat = I(GEOSRAD_LWR_FLX_RAT); U.sfcem_rat = I(GEOSRAD_LWR_SFCEM_RAT); U.dfdts_rat = I(GEOSRAD_LWR_DFDTS_RAT);
        auto O = [&](int k) { return (R *)out[k]; };
        U.dolr = O(GEOSRAD_LWR_DOLR); U.dlws = O(GEOSRAD_LWR_DLWS); U.dflns = O(GEOSRAD_LWR_DFLNS); U.dsfcem = O(GEOSRAD_LWR_DSFCEM);
        U.nettrap = O(GEOSRAD_LWR_NETTRAP); U.coltrap = O(GEOSRAD_LWR_COLTRAP); U.flx = O(GEOSRAD_LWR_FLX); U.dfdts_out = O(GEOSRAD_LWR_DFDTS_OUT);
        hipLaunchKernelGGL((k_lw_update_rats<R>), dim3((unsigned)((ncol + 255) / 256), lm + 1, nrats), dim3(256), 0, st, U);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int lw_update_bands_dev(hipStream_t st, int ncol, const int32_t *band_output, const double *wn1, const double *wn2, double undef,
                            const void *tsinst, const void *ts_int, const void *olrb_int, const void *dolrb_int, void *olrb_exp,
                            void *tbrb_exp) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || !band_output || !wn1 || !wn2 || !tsinst || !ts_int || !olrb_int || !dolrb_int) return fail(GEOSRAD_EINVAL, "bad arguments");
        if (!olrb_exp && !tbrb_exp) return GEOSRAD_OK;
        if (!d_bandflags) HIPCHK(hipMalloc((void **)&d_bandflags, 16 * sizeof(int)));
        HIPCHK(hipMemsetAsync(d_bandflags, 0, 16 * sizeof(int), st));
        LwBandUpd<R> U{};
        U.ncol = ncol; U.undef = (R)undef;
        for (int b = 0; b < 16; b++) {
            U.band_output[b] = band_output[b] != 0;
            U.wn1[b] = (R)wn1[b] * (R)100.; U.wn2[b] = (R)wn2[b] * (R)100.;      // wavenum1(ibnd)*100. [m-1] (IRR:4016)
            if (U.band_output[b] && !(wn2[b] > wn1[b] && wn1[b] + wn2[b] > 0)) return fail(GEOSRAD_EINVAL, "band limits must satisfy 0 <= wavenum1 < wavenum2");
        }
        U.tsinst = (const R *)tsinst; U.ts_int = (const R *)ts_int; U.olrb_int = (const R *)olrb_int; U.dolrb_int = (const R *)dolrb_int;
        U.olrb_exp = (R *)olrb_exp; U.tbrb_exp = (R *)tbrb_exp; U.nonzero = d_bandflags;
        const dim3 grid((unsigned)((ncol + 255) / 256), 16);
        hipLaunchKernelGGL((k_lw_update_bands<R, 0>), grid, dim3(256), 0, st, U);
        if (tbrb_exp) hipLaunchKernelGGL((k_lw_update_bands<R, 1>), grid, dim3(256), 0, st, U);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int sw_update_export_dev(hipStream_t st, int ncol, int lm, int nbands, const void *const *in, void *const *out) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || lm <= 0 || nbands < 0) return fail(GEOSRAD_EINVAL, "bad ncol/lm/nbands");
        SwUpd<R> U{};
        U.ncol = ncol; U.lm = lm; U.nbands = nbands;
        static_assert(offsetof(SwUpd<R>, fswbandnan) - offsetof(SwUpd<R>, slr) == (GEOSRAD_SWU_NIN - 1) * sizeof(void *), "SwUpd input layout");
        static_assert(offsetof(SwUpd<R>, osrcna) - offsetof(SwUpd<R>, fsw) == (GEOSRAD_SWU_NOUT - 1) * sizeof(void *), "SwUpd export layout");
        const R **ip = &U.slr;      // members in the order of the GEOSRAD_SWU_* enums
        for (int k = 0; k < GEOSRAD_SWU_NIN; k++) ip[k] = (const R *)in[k];
        R **op = &U.fsw;
        for (int k = 0; k < GEOSRAD_SWU_NOUT; k++) op[k] = (R *)out[k];
        if (!U.slr) return fail(GEOSRAD_EINVAL, "SLR is required");
        // an export needs the internals it is computed from
        auto need = [&](const R *o, const R *a, const R *b = (const R *)1) { return !o || (a && b); };
        const bool ok = need(U.fsw, U.fswn) && need(U.fsc, U.fscn) && need(U.fswna, U.fswnan) && need(U.fscna, U.fscnan) &&
                        need(U.fswu, U.fswun) && need(U.fscu, U.fscun) && need(U.fswuna, U.fswunan) && need(U.fscuna, U.fscunan) &&
                        need(U.fswd, U.fswn, U.fswun) && need(U.fscd, U.fscn, U.fscun) && need(U.fswdna, U.fswnan, U.fswunan) &&
                        need(U.fscdna, U.fscnan, U.fscunan) && need(U.fswband, U.fswbandn) && need(U.fswbandna, U.fswbandnan) &&
                        need(U.rsr, U.fswn) && need(U.rsrs, U.fswn) && need(U.osr, U.fswn) && need(U.rsc, U.fscn) && need(U.rscs, U.fscn) &&
                        need(U.osrclr, U.fscn) && need(U.rsrna, U.fswnan) && need(U.rsrsna, U.fswnan) && need(U.osrna, U.fswnan) &&
                        need(U.rscna, U.fscnan) && need(U.rscsna, U.fscnan) && need(U.osrcna, U.fscnan);
        if (!ok) return fail(GEOSRAD_EINVAL, "an export was requested without the internal field it is computed from");
        constexpr int VW = 16 / (int)sizeof(R);
        bool wide = ncol % VW == 0;
        for (int k = 0; k < GEOSRAD_SWU_NIN; k++) wide = wide && ((uintptr_t)in[k] & 15) == 0;
        for (int k = 0; k < GEOSRAD_SWU_NOUT; k++) wide = wide && ((uintptr_t)out[k] & 15) == 0;
        if (wide) hipLaunchKernelGGL((k_sw_update_export<R, VW>), dim3((unsigned)((ncol / VW + 255) / 256), lm + 1 + nbands), dim3(256), 0, st, U);
        else hipLaunchKernelGGL((k_sw_update_export<R, 1>), dim3((unsigned)((ncol + 255) / 256), lm + 1 + nbands), dim3(256), 0, st, U);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int sw_update_surface_dev(hipStream_t st, int ncol, int lm, double undef, const void *const *in, void *const *out) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || lm <= 0) return fail(GEOSRAD_EINVAL, "bad ncol/lm");
        for (int k = 0; k <= GEOSRAD_SWS_FSWN; k++) {
            const bool alb_imp = k >= GEOSRAD_SWS_ALBVF && k <= GEOSRAD_SWS_ALBNR;
            if (!in[k] && !(alb_imp && !out[GEOSRAD_SWS_ALBVF_X + (k - GEOSRAD_SWS_ALBVF)]) &&
                !(k == GEOSRAD_SWS_ZTH && !out[GEOSRAD_SWS_DRNUVR] && !out[GEOSRAD_SWS_DRNPAR] && !out[GEOSRAD_SWS_DRNNIR]))
                return fail(GEOSRAD_EINVAL, "null input field");
        }
        if ((!in[GEOSRAD_SWS_FSCN] && (out[GEOSRAD_SWS_SLRSFC] || out[GEOSRAD_SWS_SLRSUFC])) ||
            (!in[GEOSRAD_SWS_FSWNAN] && (out[GEOSRAD_SWS_SLRSFNA] || out[GEOSRAD_SWS_SLRSUFNA])) ||
            (!in[GEOSRAD_SWS_FSCNAN] && (out[GEOSRAD_SWS_SLRSFCNA] || out[GEOSRAD_SWS_SLRSUFCNA])))
            return fail(GEOSRAD_EINVAL, "a requested surface export needs an internal flux that is null");
        SwSfc<R> U{};
        U.ncol = ncol; U.lm = lm; U.undef = (R)undef;
        auto I = [&](int k) { return (const R *)in[k]; };
        auto O = [&](int k) { return (R *)out[k]; };
        U.slr = I(GEOSRAD_SWS_SLR); U.zth = I(GEOSRAD_SWS_ZTH);
        for (int k = 0; k < 4; k++) { U.alb_imp[k] = I(GEOSRAD_SWS_ALBVF + k); U.alb_exp[k] = O(GEOSRAD_SWS_ALBVF_X + k); }
        for (int k = 0; k < 6; k++) { U.dn[k] = I(GEOSRAD_SWS_DRUVRN + k); U.dx[k] = O(GEOSRAD_SWS_DRUVR + k); }
        U.fswn = I(GEOSRAD_SWS_FSWN); U.fscn = I(GEOSRAD_SWS_FSCN); U.fswnan = I(GEOSRAD_SWS_FSWNAN); U.fscnan = I(GEOSRAD_SWS_FSCNAN);
        U.albedo = O(GEOSRAD_SWS_ALBEDO); U.slrtp = O(GEOSRAD_SWS_SLRTP);
        for (int k = 0; k < 3; k++) U.drn[k] = O(GEOSRAD_SWS_DRNUVR + k);
        U.slrsf = O(GEOSRAD_SWS_SLRSF); U.slrsfc = O(GEOSRAD_SWS_SLRSFC); U.slrsfna = O(GEOSRAD_SWS_SLRSFNA); U.slrsfcna = O(GEOSRAD_SWS_SLRSFCNA);
        U.slrsuf = O(GEOSRAD_SWS_SLRSUF); U.slrsufc = O(GEOSRAD_SWS_SLRSUFC); U.slrsufna = O(GEOSRAD_SWS_SLRSUFNA); U.slrsufcna = O(GEOSRAD_SWS_SLRSUFCNA);
        hipLaunchKernelGGL((k_sw_update_surface<R>), dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, st, U);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    // ---- lit-column compaction (GEOS_SolarGridComp.F90:3686, PackIt / UnPackIt :7753-7799) --------------------------------------
    int lit_index_dev(hipStream_t st, int ncol, const void *zth, int32_t *idx, int32_t *pos, int32_t *nlit_dev, int *nlit_host) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || !zth || !idx || !pos || !nlit_dev) return fail(GEOSRAD_EINVAL, "lit_index: bad arguments");
        hipLaunchKernelGGL(k_lit_index<R>, dim3(1), dim3(1024), 0, st, ncol, (const R *)zth, idx, pos, nlit_dev);
        HIPCHK(hipGetLastError());
        if (nlit_host) {      // the caller sizes the packed call with it (NumLit = count(daytime) in the GridComp)
            int32_t v = 0;
            HIPCHK(hipMemcpyAsync(&v, nlit_dev, sizeof v, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            *nlit_host = v;
        }
        return GEOSRAD_OK;
    }
    int lit_pack_dev(hipStream_t st, int pdim, int udim, int nlev, const int32_t *idx, const int32_t *nlit_dev, const void *unpacked,
                     void *packed) override
    {
        HIPCHK(hipSetDevice(device));
        if (pdim <= 0 || udim <= 0 || nlev <= 0 || !idx || !nlit_dev || !unpacked || !packed) return fail(GEOSRAD_EINVAL, "lit_pack: bad arguments");
        const int nmax = pdim < udim ? pdim : udim;
        hipLaunchKernelGGL(k_lit_pack<R>, dim3((unsigned)((nmax + 255) / 256), nlev), dim3(256), 0, st, pdim, udim, idx, nlit_dev,
                           (const R *)unpacked, (R *)packed);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }
    int lit_unpack_dev(hipStream_t st, int pdim, int udim, int nlev, const int32_t *pos, const void *packed, void *unpacked, int use_default,
                       double dflt) override
    {
        HIPCHK(hipSetDevice(device));
        if (pdim <= 0 || udim <= 0 || nlev <= 0 || !pos || !unpacked || !packed) return fail(GEOSRAD_EINVAL, "lit_unpack: bad arguments");
        hipLaunchKernelGGL(k_lit_unpack<R>, dim3((unsigned)((udim + 255) / 256), nlev), dim3(256), 0, st, pdim, udim, pos, (const R *)packed,
                           (R *)unpacked, use_default, (R)dflt);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int rad_tendencies_dev(hipStream_t st, int ncol, int lm, double grav, double cp, const void *const *in, void *const *out) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || lm <= 0) return fail(GEOSRAD_EINVAL, "bad ncol/lm");
        RadTend<R> P{};
        P.ncol = ncol; P.lm = lm; P.grav = (R)grav; P.cp = (R)cp;
        static_assert(offsetof(RadTend<R>, trd) - offsetof(RadTend<R>, ple) == (GEOSRAD_RT_NIN - 1) * sizeof(void *), "RadTend input layout");
        static_assert(offsetof(RadTend<R>, radsrf) - offsetof(RadTend<R>, dtdt) == (GEOSRAD_RT_NOUT - 1) * sizeof(void *), "RadTend export layout");
        const R **ip = &P.ple;
        for (int k = 0; k < GEOSRAD_RT_NIN; k++) ip[k] = (const R *)in[k];
        R **op = &P.dtdt;
        for (int k = 0; k < GEOSRAD_RT_NOUT; k++) op[k] = (R *)out[k];
        auto need = [&](const R *o, const R *a, const R *b = (const R *)1, const R *c = (const R *)1) { return !o || (a && b && c); };
        const bool any3 = P.radlw || P.radsw || P.radlwc || P.radswc || P.radswna || P.radlwcna || P.radswcna;
        const bool ok = need(P.dtdt, P.flw, P.fsw) && (!any3 || P.ple) && need(P.radlw, P.flw) && need(P.radsw, P.fsw) &&
                        need(P.radlwc, P.flwclr) && need(P.radswc, P.fswclr) && need(P.radswna, P.fswna) && need(P.radlwcna, P.fla) &&
                        need(P.radswcna, P.fscna) && need(P.blw, P.dsfdts) && need(P.alw, P.sfcem, P.dsfdts, P.trd) &&
                        need(P.radsrf, P.fsw, P.flw);
        if (!ok) return fail(GEOSRAD_EINVAL, "an export was requested without the field it is computed from");
        constexpr int VW = 16 / (int)sizeof(R);
        bool wide = ncol % VW == 0;
        for (int k = 0; k < GEOSRAD_RT_NIN; k++) wide = wide && ((uintptr_t)in[k] & 15) == 0;
        for (int k = 0; k < GEOSRAD_RT_NOUT; k++) wide = wide && ((uintptr_t)out[k] & 15) == 0;
        if (wide) hipLaunchKernelGGL((k_rad_tendencies<R, VW>), dim3((unsigned)((ncol / VW + 255) / 256), lm), dim3(256), 0, st, P);
        else hipLaunchKernelGGL((k_rad_tendencies<R, 1>), dim3((unsigned)((ncol + 255) / 256), lm), dim3(256), 0, st, P);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int check(hipStream_t st, int which = -1) override
    {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamSynchronize(st));
        uint32_t e2[2] = {0, 0};
        HIPCHK(hipMemcpy(e2, d_err, 8, hipMemcpyDeviceToHost));
        // slot 0 = RRTMG_LW (+ McICA), slot 1 = RRTMG_SW; the Chou schemes have no input assertions (neither has the reference).
        // The host-pointer entry points look at their own solver's slot only (a flag left by an unchecked `_dev` call of the other
        // solver is that call's to report); geosrad_check reports either.  Only the slot that is reported is cleared - with the two
        // solvers on two streams the other one's flag stays up for the check of its own stream - and it is cleared in stream order
        // on `st`, so a kernel of the other solver running on another stream cannot lose a bit to it.
        if (which == 0) e2[1] = 0;
        if (which == 1) e2[0] = 0;
        if (!e2[0] && !e2[1]) return GEOSRAD_OK;
        HIPCHK(hipMemsetAsync(d_err + (e2[1] ? 1 : 0), 0, 4, st));
        HIPCHK(hipStreamSynchronize(st));
        if (e2[1]) {   // RRTMG_SW input assertions (SW/rrtmg_sw_rad.F90:365-383)
            static const char *SW_NEG_NAMES[12] = {"play", "tlay", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr", "cld", "ciwp", "clwp", "rei", "rel"};
            for (int k = 0; k < 12; k++)
                if (e2[1] & (1u << k)) return fail(GEOSRAD_EINPUT, std::string("negative values in input: ") + SW_NEG_NAMES[k]);
            if (e2[1] & (1u << SWERR_PLEV)) return fail(GEOSRAD_EINPUT, "negative values in input: plev");
            if (e2[1] & (1u << SWERR_ALB)) return fail(GEOSRAD_EINPUT, "negative values in input: surface albedo");
            if (e2[1] & (1u << SWERR_AER)) return fail(GEOSRAD_EINPUT, "negative values in input: aerosol optical properties");
            return fail(GEOSRAD_EINPUT, "device-side input check failed (rrtmg_sw)");
        }
        const uint32_t e = e2[0];
        for (int k = 0; k < 21; k++)
            if (e & (1u << k)) return fail(GEOSRAD_EINPUT, std::string("negative values in input: ") + LW_NEG_NAMES[k]);
        if (e & (1u << ERR_PRESSURE_ORDER)) return fail(GEOSRAD_EINPUT, "RRTMG LW pressure misordering");
        if (e & (1u << ERR_ICE_RADIUS_HI)) return fail(GEOSRAD_EINPUT, "cldprmc: iceflag: excessive high-radius extrapolation forbidden!");
        if (e & (1u << ERR_ICE_RADIUS_LO)) return fail(GEOSRAD_EINPUT, "cldprmc: iceflag: excessive low-radius extrapolation forbidden!");
        if (e & (1u << ERR_LIQ_RADIUS_HI)) return fail(GEOSRAD_EINPUT, "cldprmc: liqflag 1: excessive high-radius extrapolation forbidden!");
        if (e & (1u << ERR_LIQ_RADIUS_LO)) return fail(GEOSRAD_EINPUT, "cldprmc: liqflag 1: excessive low-radius extrapolation forbidden!");
        return fail(GEOSRAD_EINPUT, "device-side input check failed");
    }

    // host-pointer entry points start from a clean slot of their own solver (stream-ordered on the internal stream)
    int clear_slot(int which) { HIPCHK(hipMemsetAsync(d_err + which, 0, 4, stream)); return GEOSRAD_OK; }

    int ensure_io(size_t bytes)
    {
        if (bytes <= io_bytes) return GEOSRAD_OK;
        if (d_io) { HIPCHK(hipFree(d_io)); d_io = nullptr; io_bytes = 0; }
        hipError_t e = hipMalloc((void **)&d_io, bytes);
        if (e != hipSuccess) return fail(GEOSRAD_ENOMEM, "hipMalloc of the host-API staging buffer failed");
        io_bytes = bytes;
        return GEOSRAD_OK;
    }

    // ---- RRTMG_LW, host pointers --------------------------------------------------------------------------------
    int lw_host(int ncol, int nlay, int dudTs, const void *const *in, int iceflg, int liqflg, int dyofyr, int cloudLM, int cloudMH,
                int32_t *clearCounts, void *const *out, const int32_t *band_output, void *taug, void *pfracs) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || nlay <= 0) return fail(GEOSRAD_EINVAL, "bad ncol/nlay");
        if (!taug) {
            // production path: pinned staging + chunk pipeline (host_pipeline); the taug / pfracs test hook keeps the plain path below
            for (int k = 0; k < I_NIN; k++) if (!in[k] && k != I_TAUAER) return fail(GEOSRAD_EINVAL, "null input array");
            for (int k = 0; k < 4; k++) if (!out[k]) return fail(GEOSRAD_EINVAL, "null output array");
            bool any_bo = false;
            if (band_output) for (int b = 0; b < 16; b++) any_bo |= band_output[b] != 0;
            const size_t L = (size_t)nlay, E = sizeof(R);
            std::vector<PipeArr> arrs;
            int ix_in[I_NIN], ix_out[O_NOUT], ix_cc = -1;
            for (int k = 0; k < I_NIN; k++) {
                ix_in[k] = -1;
                if (!in[k]) continue;
                const size_t rows = (k == I_PLEV || k == I_TLEV) ? L + 1 : (k == I_TSFC || k == I_ALAT) ? 1 : k == I_EMIS ? 16 : k == I_TAUAER ? 16 * L : L;
                ix_in[k] = (int)arrs.size(); arrs.push_back({in[k], nullptr, rows, E, 0});
            }
            // olrb / dolrb_dTs, Fortran (16, ncol): 16 reals per column; the reference leaves un-requested bands untouched, so the
            // caller's content makes the round trip
            for (int k = 0; k < O_NOUT; k++) ix_out[k] = -1;
            if (any_bo && out[O_OLRB]) { ix_out[O_OLRB] = (int)arrs.size(); arrs.push_back({out[O_OLRB], out[O_OLRB], 1, 16 * E, 0}); }
            if (any_bo && dudTs && out[O_DOLRB]) { ix_out[O_DOLRB] = (int)arrs.size(); arrs.push_back({out[O_DOLRB], out[O_DOLRB], 1, 16 * E, 0}); }
            for (int k = 0; k < O_OLRB; k++) {
                if (!out[k] || ((k == O_DUFLX || k == O_DUFLXC) && !dudTs)) continue;
                ix_out[k] = (int)arrs.size(); arrs.push_back({nullptr, out[k], L + 1, E, 0});
            }
            ix_cc = (int)arrs.size(); arrs.push_back({nullptr, clearCounts, 4, sizeof(int32_t), 0});      // dst may be null: stays on the device
            auto run = [&](hipStream_t st, int nc, int, char *dev, int) -> int {
                const void *din[I_NIN]; void *dout[O_NOUT];
                for (int k = 0; k < I_NIN; k++) din[k] = ix_in[k] >= 0 ? dev + arrs[ix_in[k]].off : nullptr;
                for (int k = 0; k < O_NOUT; k++) dout[k] = ix_out[k] >= 0 ? dev + arrs[ix_out[k]].off : nullptr;
                return lw_dev(st, nc, nlay, dudTs, din, iceflg, liqflg, dyofyr, cloudLM, cloudMH, (int32_t *)(dev + arrs[ix_cc].off), dout,
                              band_output, nullptr, nullptr, nullptr);
            };
            int rc = clear_slot(0);
            if (rc) return rc;
            rc = host_pipeline(ncol, arrs, run, d_err);
            if (rc && rc != PIPE_FLAGGED) return rc;
            return check(stream, 0);
        }
        const size_t cl = (size_t)ncol * nlay, cv = (size_t)ncol * (nlay + 1);
        size_t insz[I_NIN];
        for (int k = 0; k < I_NIN; k++) insz[k] = cl;
        insz[I_PLEV] = insz[I_TLEV] = cv; insz[I_TSFC] = insz[I_ALAT] = ncol; insz[I_EMIS] = (size_t)ncol * 16;
        insz[I_TAUAER] = cl * 16;
        size_t outsz[O_NOUT] = {cv, cv, cv, cv, cv, cv, (size_t)ncol * 16, (size_t)ncol * 16};
        size_t off = 0;
        auto take = [&](size_t nreal) { size_t o = off; off += al(nreal * sizeof(R)); return o; };
        size_t ino[I_NIN], outo[O_NOUT];
        for (int k = 0; k < I_NIN; k++) ino[k] = in[k] ? take(insz[k]) : (size_t)-1;
        for (int k = 0; k < O_NOUT; k++) outo[k] = take(outsz[k]);
        const size_t cco = take((size_t)ncol * 4 * sizeof(int32_t) / sizeof(R) + 4);
        size_t dbgo[2] = {0, 0};
        if (taug) { dbgo[0] = take(cl * NG_LW); dbgo[1] = take(cl * NG_LW); }
        int rc = ensure_io(off);
        if (rc) return rc;
        const void *din[I_NIN]; void *dout[O_NOUT];
        for (int k = 0; k < I_NIN; k++) {
            din[k] = in[k] ? d_io + ino[k] : nullptr;
            if (in[k]) HIPCHK(hipMemcpyAsync(d_io + ino[k], in[k], insz[k] * sizeof(R), hipMemcpyHostToDevice, stream));
        }
        for (int k = 0; k < O_NOUT; k++) dout[k] = d_io + outo[k];
        if (band_output) {   // the reference leaves un-requested bands untouched: round-trip the caller's content
            bool any = false;
            for (int b = 0; b < 16; b++) any |= band_output[b] != 0;
            if (any && out[O_OLRB]) HIPCHK(hipMemcpyAsync(dout[O_OLRB], out[O_OLRB], outsz[O_OLRB] * sizeof(R), hipMemcpyHostToDevice, stream));
            if (any && dudTs && out[O_DOLRB]) HIPCHK(hipMemcpyAsync(dout[O_DOLRB], out[O_DOLRB], outsz[O_DOLRB] * sizeof(R), hipMemcpyHostToDevice, stream));
        }
        void *dout_eff[O_NOUT];
        for (int k = 0; k < O_NOUT; k++) dout_eff[k] = out[k] ? dout[k] : nullptr;
        rc = lw_dev(stream, ncol, nlay, dudTs, din, iceflg, liqflg, dyofyr, cloudLM, cloudMH, (int32_t *)(d_io + cco), dout_eff,
                    band_output, taug ? d_io + dbgo[0] : nullptr, taug ? d_io + dbgo[1] : nullptr, nullptr);
        if (rc) return rc;
        rc = check(stream, 0);
        if (rc) return rc;
        bool any_bo = false;
        if (band_output) for (int b = 0; b < 16; b++) any_bo |= band_output[b] != 0;
        for (int k = 0; k < O_NOUT; k++) {
            if (!out[k]) continue;
            if ((k == O_DUFLX || k == O_DUFLXC) && !dudTs) continue;
            if (k == O_OLRB && !any_bo) continue;
            if (k == O_DOLRB && !(any_bo && dudTs)) continue;
            HIPCHK(hipMemcpyAsync(out[k], dout[k], outsz[k] * sizeof(R), hipMemcpyDeviceToHost, stream));
        }
        if (clearCounts) HIPCHK(hipMemcpyAsync(clearCounts, d_io + cco, (size_t)ncol * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        if (taug) {
            HIPCHK(hipMemcpyAsync(taug, d_io + dbgo[0], cl * NG_LW * sizeof(R), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(pfracs, d_io + dbgo[1], cl * NG_LW * sizeof(R), hipMemcpyDeviceToHost, stream));
        }
        HIPCHK(hipStreamSynchronize(stream));
        return GEOSRAD_OK;
    }


    // =====================================================================================================
    // RRTMG_SW
    // =====================================================================================================
    int set_tables_sw(const void *blob, size_t nbytes) override
    {
        HIPCHK(hipSetDevice(device));
        Blob B;
        if (!B.parse(blob, nbytes)) return fail(GEOSRAD_ETABLE, B.err);
        if (B.realbytes != (int)sizeof(R))
            return fail(GEOSRAD_ETABLE, "table blob real size does not match the context's real_kind");
        TableStage<R> S(B);
        SwDev<R> &T = h_S;
        memset(&T, 0, sizeof(T));
        static const int ng[15] = {0, 6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12};
        static const int rowsa[15] = {0, 585, 585, 585, 585, 65, 585, 585, 65, 585, 65, 0, 65, 585, 65};
        static const int rowsb[15] = {0, 235, 1175, 235, 235, 235, 1175, 235, 0, 235, 0, 0, 235, 1175, 235};
        static const int nfor[15] = {0, 3, 4, 3, 3, 4, 4, 3, 3, 3, 0, 0, 0, 0, 4};
        static const int nsrc[15] = {0, 1, 5, 9, 9, 1, 9, 9, 1, 9, 1, 1, 1, 5, 1};
        char nm[64];
        for (int b = 1; b <= NB_SW; b++) {
            SwBandTab<R> &bt = T.b[b];
            auto N = [&](const char *s) { snprintf(nm, sizeof nm, "b%02d_%s", b + 15, s); return std::string(nm); };
            if (rowsa[b]) S.tr2(&bt.absa, N("absa"), rowsa[b], ng[b]);
            if (rowsb[b]) S.tr2(&bt.absb, N("absb"), rowsb[b], ng[b]);
            if (nfor[b]) { S.tr2(&bt.selfref, N("selfref"), 10, ng[b]); S.tr2(&bt.forref, N("forref"), nfor[b], ng[b]); }
            S.rows(&bt.sflux, N("sfluxref"), ng[b], nsrc[b]); S.rows(&bt.irrad, N("irradnce"), ng[b], nsrc[b]);
            S.rows(&bt.facb, N("facbrght"), ng[b], nsrc[b]); S.rows(&bt.snsp, N("snsptdrk"), ng[b], nsrc[b]);
            const int jb = b + 15;
            if (jb == 24) { S.rows(&bt.rayl, N("rayla"), ng[b], 9); S.rows(&bt.raylb, N("raylb"), ng[b], 1); }
            else if (jb == 23 || jb == 25 || jb == 26 || jb == 27) S.rows(&bt.rayl, N("rayl"), ng[b], 1);
            else S.splat(&bt.rayl, N("rayl"), ng[b]);
            if (jb == 20) S.rows(&bt.x0, N("absch4"), ng[b], 1);
            if (jb == 24 || jb == 25) { S.rows(&bt.x0, N("abso3a"), ng[b], 1); S.rows(&bt.x1, N("abso3b"), ng[b], 1); }
            if (jb == 29) { S.rows(&bt.x0, N("absco2"), ng[b], 1); S.rows(&bt.x1, N("absh2o"), ng[b], 1); }
        }
        S.raw(&T.preflog, "preflog", 59); S.raw(&T.tref, "tref", 59);
        S.raw(&T.extliq1, "extliq1", 58 * 14); S.raw(&T.ssaliq1, "ssaliq1", 58 * 14); S.raw(&T.asyliq1, "asyliq1", 58 * 14);
        S.raw(&T.extice2, "extice2", 43 * 14); S.raw(&T.ssaice2, "ssaice2", 43 * 14); S.raw(&T.asyice2, "asyice2", 43 * 14);
        S.raw(&T.extice3, "extice3", 46 * 14); S.raw(&T.ssaice3, "ssaice3", 46 * 14); S.raw(&T.asyice3, "asyice3", 46 * 14);
        S.raw(&T.fdlice3, "fdlice3", 46 * 14);
        S.raw(&T.extice4, "extice4", 200 * 14); S.raw(&T.ssaice4, "ssaice4", 200 * 14); S.raw(&T.asyice4, "asyice4", 200 * 14);
        {
            const char *bn[6] = {"abari", "bbari", "cbari", "dbari", "ebari", "fbari"};
            R *dst[6] = {T.abari, T.bbari, T.cbari, T.dbari, T.ebari, T.fbari};
            for (int k = 0; k < 6; k++) { const R *s = S.get(bn[k], 5); if (s) memcpy(dst[k], s, 5 * sizeof(R)); }
        }
        T.oneminus = S.scalar("oneminus"); T.grav = S.scalar("grav"); T.avogad = S.scalar("avogad"); T.rrsw_scon = S.scalar("rrsw_scon");
        T.Iint = S.scalar("Iint"); T.Fint = S.scalar("Fint"); T.Sint = S.scalar("Sint");
        T.Mg_avg = S.scalar("Mg_avg"); T.Mg_0 = S.scalar("Mg_0"); T.SB_avg = S.scalar("SB_avg"); T.SB_0 = S.scalar("SB_0");
        {   // AvgCyc11 of the two indices (isolvar == 1 only; blobs written before round 4 lack them: that option is then refused)
            avgcyc_mg.clear(); avgcyc_sb.clear();
            if (B.e.count("mgavgcyc") && B.e.count("sbavgcyc")) {
                const R *m = S.get("mgavgcyc", 134), *q = S.get("sbavgcyc", 134);
                if (m && q) { avgcyc_mg.assign(m, m + 134); avgcyc_sb.assign(q, q + 134); }
            }
        }
        {
            int32_t icxa[14], ngb[112];
            if (S.ints("icxa", 14, icxa)) for (int b = 1; b <= NB_SW; b++) T.icxa[b] = icxa[b - 1];
            // the band <-> g-point map is compiled into the kernels; refuse tables that disagree
            if (S.ints("ngb", 112, ngb)) {
                int g = 0;
                for (int b = 1; b <= NB_SW; b++) for (int k = 0; k < ng[b]; k++, g++) if (ngb[g] != b + 15) S.missing += "ngb(mismatch) ";
            }
        }
        if (!S.missing.empty()) return fail(GEOSRAD_ETABLE, "missing/ill-shaped table entries: " + S.missing);
        if (d_tab_sw) { HIPCHK(hipFree(d_tab_sw)); d_tab_sw = nullptr; }
        tab_sw_bytes = S.stage.size();
        HIPCHK(hipMalloc((void **)&d_tab_sw, tab_sw_bytes));
        HIPCHK(hipMemcpy(d_tab_sw, S.stage.data(), tab_sw_bytes, hipMemcpyHostToDevice));
        for (auto &f : S.fix) *f.first = (const R *)(d_tab_sw + f.second);
        // sw_eval reaches a band's upper-atmosphere tables as the lower ones' base + a 32-bit byte offset (one allocation, staged in this order)
        for (int b = 1; b <= NB_SW; b++)
            if ((T.b[b].absb && T.b[b].absb < T.b[b].absa) || (T.b[b].x1 && T.b[b].x1 < T.b[b].x0) || tab_sw_bytes >= ((size_t)1 << 32))
                return fail(GEOSRAD_ETABLE, "internal: RRTMG_SW table staging order");
        HIPCHK(hipMemcpy(d_S, &h_S, sizeof(SwDev<R>), hipMemcpyHostToDevice));
        have_sw = true;
        return GEOSRAD_OK;
    }

    struct WsSw { R *sc; uint32_t *scidx; uint8_t *colcloudy, *laycloudy; int32_t *perm, *nclear; R *alpha, *rcorr, *taucmc, *ssacmc, *asmcmc, *cotsum, *cell, *part, *bsfc, *cot; };
    size_t ws_layout_sw(int nc, int nlay, WsSw *w, char *base, int planes) const
    {
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return base ? base + o : (char *)nullptr; };
        const size_t cl = (size_t)nlay * nc;
        char *p;
        p = take(SW_NFIELD * cl * sizeof(R)); if (w) w->sc = (R *)p;
        p = take(cl * 4); if (w) w->scidx = (uint32_t *)p;
        p = take(nc); if (w) w->colcloudy = (uint8_t *)p;
        p = take((size_t)nc * 4); if (w) w->perm = (int32_t *)p;
        p = take(4); if (w) w->nclear = (int32_t *)p;
        p = take(cl); if (w) w->laycloudy = (uint8_t *)p;
        p = take(cl * sizeof(R)); if (w) w->alpha = (R *)p;
        p = take(cl * sizeof(R)); if (w) w->rcorr = (R *)p;
        p = take(NG_SW * cl * sizeof(R)); if (w) w->taucmc = (R *)p;
        p = take(NG_SW * cl * sizeof(R)); if (w) w->ssacmc = (R *)p;
        p = take(NG_SW * cl * sizeof(R)); if (w) w->asmcmc = (R *)p;
        p = take((size_t)3 * NG_SW * nc * sizeof(R)); if (w) w->cotsum = (R *)p;
        // parked planes of the band sweeps: k_sw_reform fp32 5 (gas optical depth + 2 x 2 upward reflectances), fp64 15 (no gas optical
        // depth, 2 x 5 layer properties instead); k_sw_bands 14
        // (the stage-dump instantiation is always k_sw_bands: sw_planes(true))
        const size_t nplanes = (size_t)planes;
        p = take(nplanes * NG_SW * nlay * (((size_t)nc + 255) & ~(size_t)255) * sizeof(R)); if (w) w->cell = (R *)p;
        // partial fluxes per slot: the units of k_sw_reform's mapping (23 fp32 / 32 fp64), which also cover the 14 bands of k_sw_bands (the
        // stage-dump hook always runs that kernel); 14 when GEOSRAD_SW_PATH=bands
        const size_t slots = sw_path == 2 ? (size_t)(sw_reform_nslot<R>() > NB_SW ? sw_reform_nslot<R>() : NB_SW) : (size_t)NB_SW;
        p = take((size_t)4 * slots * (nlay + 1) * nc * sizeof(R)); if (w) w->part = (R *)p;
        p = take((size_t)3 * slots * nc * sizeof(R)); if (w) w->bsfc = (R *)p;
        p = take((size_t)8 * 6 * nc * sizeof(R)); if (w) w->cot = (R *)p;
        return off;
    }
    int sw_planes(bool dbg) const { return (sw_path == 2 && !dbg) ? (sizeof(R) == 4 ? 5 : 15) : 14; }
    int ensure_ws_sw(int nc, int nlay, int planes)
    {
        if (d_ws_sw && nc <= ws_sw_ncol && nlay == ws_sw_nlay && planes <= ws_sw_planes) return GEOSRAD_OK;
        if (planes < ws_sw_planes) planes = ws_sw_planes;
        const int want = (d_ws_sw && nlay == ws_sw_nlay && nc < ws_sw_ncol) ? ws_sw_ncol : nc;
        if (d_ws_sw) { HIPCHK(hipFree(d_ws_sw)); d_ws_sw = nullptr; ws_sw_bytes = 0; }
        const size_t need = ws_layout_sw(want, nlay, nullptr, nullptr, planes);
        hipError_t e = hipMalloc((void **)&d_ws_sw, need);
        if (e != hipSuccess) return fail(GEOSRAD_ENOMEM, "hipMalloc of the SW workspace failed (" + std::to_string(need >> 20) +
                                                         " MiB); lower it with geosrad_set_chunk()");
        ws_sw_bytes = need; ws_sw_ncol = want; ws_sw_nlay = nlay; ws_sw_planes = planes;
        return GEOSRAD_OK;
    }

    // solar variability block of the driver (SW/rrtmg_sw_rad.F90:893-1127), scalars only
    // NRLSSI2's host routines for isolvar == 1 (SW/NRLSSI2.F90; nsolfrac = 134, intrvl_len = 1 / 132) - false where the reference error-stops
    static bool nrl_adjust(R solcycfr, const R *indsolvar, R *scl)          // adjust_solcyc_amplitudes (:236-271)
    {
        const R fmin = (R)0.0189, fmax = (R)0.3750, dmin2max = fmax - fmin, dmax2min = (R)1. - dmin2max;
        if (solcycfr >= 0 && solcycfr < fmin) {
            const R wgt = (solcycfr + (R)1. - fmax) / dmax2min;
            scl[0] = indsolvar[0] + wgt * ((R)1. - indsolvar[0]); scl[1] = indsolvar[1] + wgt * ((R)1. - indsolvar[1]);
        } else if (solcycfr >= fmin && solcycfr <= fmax) {
            const R wgt = (solcycfr - fmin) / dmin2max;
            scl[0] = (R)1. + wgt * (indsolvar[0] - (R)1.); scl[1] = (R)1. + wgt * (indsolvar[1] - (R)1.);
        } else if (solcycfr > fmax && solcycfr <= 1) {
            const R wgt = (solcycfr - fmax) / dmax2min;
            scl[0] = indsolvar[0] + wgt * ((R)1. - indsolvar[0]); scl[1] = indsolvar[1] + wgt * ((R)1. - indsolvar[1]);
        } else return false;
        return true;
    }
    bool nrl_interp(R solcycfr, R &Mg, R &SB) const                         // interpolate_indices (:277-332)
    {
        const R il = (R)1.0 / (R)132, ilh = (R)0.5 * il;
        const R *mg = avgcyc_mg.data() - 1, *sb = avgcyc_sb.data() - 1;     // 1-based like the reference
        if (solcycfr > 0 && solcycfr < 1) {
            int sfid; R lo, hi;
            if (solcycfr <= ilh) { sfid = 1; lo = 0; hi = ilh; }
            else if (solcycfr > ilh && solcycfr < (R)1. - ilh) { sfid = (int)std::floor((solcycfr - ilh) * (R)132) + 2; lo = (R)(sfid - 2) * il + ilh; hi = lo + il; }
            else { sfid = 133; lo = (R)1. - ilh; hi = 1; }
            const R f = (solcycfr - lo) / (hi - lo);
            Mg = mg[sfid] + f * (mg[sfid + 1] - mg[sfid]); SB = sb[sfid] + f * (sb[sfid + 1] - sb[sfid]);
        } else if (solcycfr == 0) { Mg = mg[1]; SB = sb[1]; }
        else if (solcycfr == 1) { Mg = mg[134]; SB = sb[134]; }
        else return false;
        return true;
    }
    void nrl_means(const R *ind_opt, R &mean_f, R &mean_s) const             // initialize_NRLSSI2, isolvar == 1 (:160-232)
    {
        const SwDev<R> &T = h_S;
        const R il = (R)1.0 / (R)132, ilh = (R)0.5 * il;
        const R *mg = avgcyc_mg.data() - 1, *sb = avgcyc_sb.data() - 1;
        const R ind[2] = {ind_opt ? ind_opt[0] : (R)1, ind_opt ? ind_opt[1] : (R)1};
        mean_f = 1; mean_s = 1;
        const bool s1 = ind[0] != 1, s2 = ind[1] != 1;
        if (!s1 && !s2) return;
        const R m1 = ((R)1. + ind[0]) / (R)2., m2 = ((R)1. + ind[1]) / (R)2.;
        R a1 = 0, a2 = 0, scl[2], fr = ilh;
        for (int n = 2; n <= 133; n++) {
            nrl_adjust(fr, ind, scl);
            if (s1) a1 = a1 + scl[0] * mg[n];
            if (s2) a2 = a2 + scl[1] * sb[n];
            fr = fr + il;
        }
        if (s1) { a1 = a1 / (R)132; mean_f = (a1 - m1 * T.Mg_0) / (T.Mg_avg - T.Mg_0); }
        if (s2) { a2 = a2 / (R)132; mean_s = (a2 - m2 * T.SB_0) / (T.SB_avg - T.SB_0); }
    }

    int sw_solar(double scon_d, double adjes_d, int isolvar, const R *bndscl, const R *indsolvar, const R *solcycfrac, SwSolar<R> &SV)
    {
        const SwDev<R> &T = h_S;
        const R scon = (R)scon_d, adjes = (R)adjes_d;
        R solvar[NB_SW + 1];
        for (int b = 0; b <= NB_SW; b++) { solvar[b] = 1; SV.svar_bnd[b] = 1; SV.adjflux[b] = 1; }
        SV.isolvar = isolvar; SV.svar_f = 1; SV.svar_s = 1; SV.svar_i = 1;
        if (isolvar != -1 && isolvar != 0 && isolvar != 1 && isolvar != 2 && isolvar != 3) return fail(GEOSRAD_EINPUT, "invalid isolvar");
        if (isolvar == 1) {
            // position in AvgCyc11 from solcycfrac, amplitude scaling from indsolvar (rrtmg_sw_rad.F90:906-930, :994-1008, :1060-1079)
            if (!solcycfrac) return fail(GEOSRAD_EINPUT, "isolvar == 1 requires solcycfrac present!");
            if (avgcyc_mg.size() != 134 || avgcyc_sb.size() != 134)
                return fail(GEOSRAD_ETABLE, "isolvar == 1 needs the mgavgcyc / sbavgcyc entries of the RRTMG_SW table blob");
            if (scon < 0) return fail(GEOSRAD_EINPUT, "scon must be >= 0");
            const R fr = *solcycfrac;
            R scl[2] = {1, 1}, Mg_now, SB_now, mean_f, mean_s;
            if (indsolvar && (indsolvar[0] != 1 || indsolvar[1] != 1))
                if (!nrl_adjust(fr, indsolvar, scl)) return fail(GEOSRAD_EINPUT, "RRTMG_SW: solcycfr must be in [0,1]");
            nrl_means(indsolvar, mean_f, mean_s);
            if (!nrl_interp(fr, Mg_now, SB_now)) return fail(GEOSRAD_EINPUT, "RRTMG_SW: solcycfr must be in [0,1]");
            SV.svar_f = scl[0] * (Mg_now - T.Mg_0) / (T.Mg_avg - T.Mg_0);
            SV.svar_s = scl[1] * (SB_now - T.SB_0) / (T.SB_avg - T.SB_0);
            SV.svar_i = scon == 0 ? (R)1 : (scon - (mean_f * T.Fint + mean_s * T.Sint)) / T.Iint;
        }
        R ndx0 = T.Mg_avg, ndx1 = T.SB_avg;
        if (isolvar == 2 && indsolvar) { ndx0 = indsolvar[0]; ndx1 = indsolvar[1]; }
        if (scon == 0) {
            if (isolvar == -1) { if (bndscl) for (int b = 1; b <= NB_SW; b++) solvar[b] = bndscl[b - 1]; }
            else if (isolvar == 2) { SV.svar_f = (ndx0 - T.Mg_0) / (T.Mg_avg - T.Mg_0); SV.svar_s = (ndx1 - T.SB_0) / (T.SB_avg - T.SB_0); SV.svar_i = 1; }
            else if (isolvar == 3) { if (bndscl) for (int b = 1; b <= NB_SW; b++) solvar[b] = bndscl[b - 1];
                for (int b = 1; b <= NB_SW; b++) SV.svar_bnd[b] = solvar[b]; }
        } else if (scon > 0) {
            const R scon_int = T.Fint + T.Sint + T.Iint;
            if (isolvar == -1) { for (int b = 1; b <= NB_SW; b++) solvar[b] = scon / T.rrsw_scon;
                if (bndscl) for (int b = 1; b <= NB_SW; b++) solvar[b] = solvar[b] * bndscl[b - 1]; }
            else if (isolvar == 0) { const R r = scon / scon_int; SV.svar_f = r; SV.svar_s = r; SV.svar_i = r; }
            else if (isolvar == 2) { SV.svar_f = (ndx0 - T.Mg_0) / (T.Mg_avg - T.Mg_0); SV.svar_s = (ndx1 - T.SB_0) / (T.SB_avg - T.SB_0);
                SV.svar_i = (scon - (SV.svar_f * T.Fint + SV.svar_s * T.Sint)) / T.Iint; }
            else { for (int b = 1; b <= NB_SW; b++) solvar[b] = scon / scon_int;
                if (bndscl) for (int b = 1; b <= NB_SW; b++) solvar[b] = solvar[b] * bndscl[b - 1];
                for (int b = 1; b <= NB_SW; b++) SV.svar_bnd[b] = solvar[b]; }
        } else return fail(GEOSRAD_EINPUT, "scon must be >= 0");
        for (int b = 1; b <= NB_SW; b++) SV.adjflux[b] = adjes;
        if (isolvar < 0) for (int b = 1; b <= NB_SW; b++) SV.adjflux[b] = SV.adjflux[b] * solvar[b];
        return GEOSRAD_OK;
    }

    // ---- RRTMG_SW, device pointers -------------------------------------------------------------------------
    int sw_dev(hipStream_t st, int ncol, int nlay, double scon, double adjes, int isolvar, const void *const *in, int iceflg, int liqflg,
               int dyofyr, int iaer, int cloudLM, int cloudMH, int normFlx, int32_t *clearCounts, void *const *out, int do_drfband,
               const void *bndscl, const void *indsolvar, const void *solcycfrac, void *const *dbg) override
    {
        return sw_run(st, ncol, nlay, scon, adjes, isolvar, in, iceflg, liqflg, dyofyr, iaer, cloudLM, cloudMH, normFlx, clearCounts, out,
                      do_drfband, bndscl, indsolvar, solcycfrac, dbg, nullptr);
    }

    // RRTMG_SW band sweeps: k_sw_reform (lane = (column, unit of g-points); fp32 re-forms the cell optics in its second sweep and parks 12
    // bytes per cell; in fp64 the second two-stream - IEEE divisions, double-precision exp / sqrt - costs more than the parked bytes it
    // would save, 32.5 against 29.4 ms per 97 200 columns, so that instantiation parks the layer properties too);
    // GEOSRAD_SW_PATH=bands selects the first mapping, k_sw_bands
    bool sw_reform_on() const { return sw_path == 2; }

    // sw_na_out (SwOutIx order, all of SO_UFLX .. SO_COT0 + 7 non-null) requests an additional pass without the aerosol terms
    int sw_run(hipStream_t st, int ncol, int nlay, double scon, double adjes, int isolvar, const void *const *in, int iceflg, int liqflg,
               int dyofyr, int iaer, int cloudLM, int cloudMH, int normFlx, int32_t *clearCounts, void *const *out, int do_drfband,
               const void *bndscl, const void *indsolvar, const void *solcycfrac, void *const *dbg, void *const *sw_na_out)
    {
        HIPCHK(hipSetDevice(device));
        if (!have_sw) return fail(GEOSRAD_EINVAL, "RRTMG_SW tables not set: call geosrad_set_tables_sw first (rrtmg_sw_ini)");
        if (ncol <= 0 || nlay < 4 || nlay > 203) return fail(GEOSRAD_EINVAL, "bad ncol/nlay (4 <= nlay <= mxlay = 203)");
        if (iceflg < 1 || iceflg > 4) return fail(GEOSRAD_EINPUT, "cldprmc_sw: invalid iceflag");
        if (liqflg != 1) return fail(GEOSRAD_EINPUT, "cldprmc_sw: invalid liqflag");
        if (cloudLM == cloudMH) return fail(GEOSRAD_EINPUT, "invalid pressure super-layers!");
        if (iaer != 0 && iaer != 10) return fail(GEOSRAD_EINPUT, "iaer must be 0 or 10");
        for (int k = 0; k < S_NIN; k++)
            if (!in[k] && !((k == S_TAUAER || k == S_SSAAER || k == S_ASMAER) && iaer != 10)) return fail(GEOSRAD_EINVAL, "null input array");
        for (int k = 0; k < SO_DRBAND; k++) if (!out[k]) return fail(GEOSRAD_EINVAL, "null output array");
        if (do_drfband && (!out[SO_DRBAND] || !out[SO_DFBAND])) return fail(GEOSRAD_EINVAL, "do_drfband set but drband/dfband null");
        SwSolar<R> SV;
        int rc = sw_solar(scon, adjes, isolvar, (const R *)bndscl, (const R *)indsolvar, (const R *)solcycfrac, SV);
        if (rc) return rc;

        // one band's [layer][g<=12][column] plane must stay below 4 GiB (32-bit byte offsets)
        const long cap = (long)(0xFFFFFFFFull / ((unsigned long long)nlay * 12ull * sizeof(R))) & ~255L;
        int nc_max = ncol < chunk ? ncol : chunk;
        if ((long)nc_max > cap) nc_max = (int)cap;
        rc = ensure_ws_sw(nc_max, nlay, sw_planes(dbg != nullptr));
        if (rc) return rc;

        for (int c0 = 0; c0 < ncol; c0 += nc_max) {
            const int nc = (ncol - c0) < nc_max ? (ncol - c0) : nc_max;
            WsSw w;
            ws_layout_sw(nc, nlay, &w, d_ws_sw, ws_sw_planes);
            SwArgs<R> A{};
            A.ncol = nc; A.ld = ncol; A.nlay = nlay; A.iceflg = iceflg; A.liqflg = liqflg; A.doy = dyofyr; A.cloudLM = cloudLM;
            A.cloudMH = cloudMH; A.iaer = iaer; A.normFlx = normFlx; A.do_drfband = do_drfband;
            auto P = [&](int k) { return in[k] ? (const R *)in[k] + c0 : (const R *)nullptr; };
            A.play = P(S_PLAY); A.plev = P(S_PLEV); A.tlay = P(S_TLAY); A.h2o = P(S_H2O); A.o3 = P(S_O3); A.co2 = P(S_CO2);
            A.ch4 = P(S_CH4); A.o2 = P(S_O2); A.cld = P(S_CLD); A.ciwp = P(S_CIWP); A.clwp = P(S_CLWP); A.rei = P(S_REI);
            A.rel = P(S_REL); A.zm = P(S_ZM); A.alat = P(S_ALAT);
            A.tauaer = iaer == 10 ? P(S_TAUAER) : nullptr; A.ssaaer = iaer == 10 ? P(S_SSAAER) : nullptr;
            A.asmaer = iaer == 10 ? P(S_ASMAER) : nullptr;
            A.coszen = P(S_COSZEN); A.asdir = P(S_ASDIR); A.asdif = P(S_ASDIF); A.aldir = P(S_ALDIR); A.aldif = P(S_ALDIF);
            A.sc = w.sc; A.scidx = w.scidx; A.colcloudy = w.colcloudy; A.laycloudy = w.laycloudy; A.perm = w.perm; A.nclear = w.nclear; A.alpha = w.alpha; A.rcorr = w.rcorr;
            A.taucmc = w.taucmc; A.ssacmc = w.ssacmc; A.asmcmc = w.asmcmc; A.cotsum = w.cotsum; A.cell = w.cell; A.part = w.part;
            A.bsfc = w.bsfc; A.cot = w.cot;
            A.err = d_err + 1;
            A.clearCounts = clearCounts + c0;
            if (dbg) {
                A.dbg_taug = (R *)dbg[0] + (size_t)c0 * NG_SW * nlay; A.dbg_taur = (R *)dbg[1] + (size_t)c0 * NG_SW * nlay;
                A.dbg_ssi = (R *)dbg[2] + (size_t)c0 * NG_SW;
            }
            const dim3 blk(256);
            const unsigned gx = (unsigned)((nc + 255) / 256);
            span_begin(6, st); hipLaunchKernelGGL(k_sw_validate<R>, dim3(gx), blk, 0, st, A);
            if (iaer == 10) hipLaunchKernelGGL(k_sw_validate_aer<R>, dim3(gx, nlay), blk, 0, st, A);
            hipLaunchKernelGGL(k_partition, dim3(1), dim3(1024), 0, st, nc, (const uint8_t *)w.colcloudy, w.perm, w.nclear); span_end(st);
            span_begin(7, st); hipLaunchKernelGGL(k_sw_setcoef<R>, dim3(gx, nlay), blk, 0, st, A, (const SwDev<R> *)d_S); span_end(st);
            span_begin(2, st); hipLaunchKernelGGL(k_overlap<R>, dim3(gx, nlay), blk, 0, st, nc, ncol, nlay, dyofyr, A.zm, A.alat,
                               (const int32_t *)w.perm, (const int32_t *)w.nclear, (const LwDev<R> *)d_T, A.alpha, A.rcorr, A.laycloudy); span_end(st);
            McArgs<R> M{};
            M.ncol = nc; M.ld = ncol; M.nlay = nlay; M.nsubcol = NG_SW; M.doy = dyofyr; M.cloudLM = cloudLM; M.cloudMH = cloudMH;
            M.iceflg = iceflg; M.liqflg = liqflg;
            M.so[0] = 4; M.so[1] = 3; M.so[2] = 2; M.so[3] = 1;        // seed_order=[4,3,2,1] (SW/rrtmg_sw_rad.F90:1401)
            M.cwp_tiny = (R)1.e-20;
            M.play = A.play; M.cldf = A.cld; M.ciwp = A.ciwp; M.clwp = A.clwp; M.rei = A.rei; M.rel = A.rel;
            M.alpha = A.alpha; M.rcorr = A.rcorr; M.perm = w.perm; M.nclear = w.nclear; M.cftop = w.colcloudy;
            M.taucmc = A.taucmc; M.ssacmc = A.ssacmc; M.asmcmc = A.asmcmc; M.laycloudy = A.laycloudy; M.cotsum = A.cotsum; M.clearCounts = A.clearCounts; M.err = d_err + 1;
            {
                McPlan MP; int nseg = 0;
                rc = mc_plan(2, NG_SW, nlay, MP, nseg);
                if (rc) return rc;
                span_begin(3, st);
                hipLaunchKernelGGL((k_mcica<R, 2>), dim3(xcd_grid(nc, 64, nseg)), dim3(64), 0, st, M, MP, (const LwDev<R> *)d_T,
                                   (const SwDev<R> *)d_S);
                span_end(st);
            }
            if (dbg && dbg[3])       // cldprmc_sw stage dump (6-entry dbg of geosrad_rrtmg_sw_cldprmc)
                hipLaunchKernelGGL(k_sw_dump_cldprmc<R>, dim3(gx, nlay), blk, 0, st, A, (R *)dbg[3] + (size_t)c0 * NG_SW * nlay,
                                   (R *)dbg[4] + (size_t)c0 * NG_SW * nlay, (R *)dbg[5] + (size_t)c0 * NG_SW * nlay);
            span_begin(8, st);
            if (dbg) {
                hipLaunchKernelGGL((k_sw_bands<R, true, true>), dim3(band_grid(nc, NB_SW)), blk, 0, st, A, h_S, SV);
            } else if (sw_reform_on()) {
                hipError_t e = sw_reform_launch<R>(st, A, h_S, SV);
                if (e != hipSuccess) return fail(GEOSRAD_EHIP, std::string("sw_reform_launch: ") + hipGetErrorString(e));
            } else {
                hipLaunchKernelGGL((k_sw_bands<R, false, false>), dim3(band_grid(nc, NB_SW)), blk, 0, st, A, h_S, SV);
                hipLaunchKernelGGL((k_sw_bands<R, true, false>), dim3(band_grid(nc, NB_SW)), blk, 0, st, A, h_S, SV);
            }
            span_end(st);
            SwOut<R> O{};
            auto Q = [&](int k) { return out[k] ? (R *)out[k] + c0 : (R *)nullptr; };
            O.swuflx = Q(SO_UFLX); O.swdflx = Q(SO_DFLX); O.swuflxc = Q(SO_UFLXC); O.swdflxc = Q(SO_DFLXC);
            O.nirr = Q(SO_NIRR); O.nirf = Q(SO_NIRF); O.parr = Q(SO_PARR); O.parf = Q(SO_PARF); O.uvrr = Q(SO_UVRR); O.uvrf = Q(SO_UVRF);
            O.fswband = Q(SO_FSWBAND);
            for (int k = 0; k < 8; k++) O.cot[k] = Q(SO_COT0 + k);
            O.drband = Q(SO_DRBAND); O.dfband = Q(SO_DFBAND);
            span_begin(9, st);
            if (sw_reform_on() && !dbg) { hipError_t e = sw_reform_reduce<R>(st, A, O); if (e != hipSuccess) return fail(GEOSRAD_EHIP, std::string("sw_reform_reduce: ") + hipGetErrorString(e)); }
            else hipLaunchKernelGGL(k_sw_reduce<R>, dim3(gx, nlay + 2), blk, 0, st, A, O);
            span_end(st);
            if (sw_na_out && !dbg) {
                // the GridComp's "no-aerosol" diagnostics (GEOS_SolarGridComp.F90:3249-3259 calls the whole of SORADCORE a second
                // time): same columns, same clouds (McICA is seeded by the pressures), same gas optical depths - only the band
                // sweeps and the reduction are repeated, without the aerosol terms; validation, setcoef and McICA are shared
                A.iaer = 0; A.do_drfband = 0;
                span_begin(8, st);
                if (sw_reform_on()) {
                    hipError_t e = sw_reform_launch<R>(st, A, h_S, SV);
                    if (e != hipSuccess) return fail(GEOSRAD_EHIP, std::string("sw_reform_launch: ") + hipGetErrorString(e));
                } else {
                    hipLaunchKernelGGL((k_sw_bands<R, false, false>), dim3(band_grid(nc, NB_SW)), blk, 0, st, A, h_S, SV);
                    hipLaunchKernelGGL((k_sw_bands<R, true, false>), dim3(band_grid(nc, NB_SW)), blk, 0, st, A, h_S, SV);
                }
                span_end(st);
                SwOut<R> N{};
                auto QN = [&](int k) { return (R *)sw_na_out[k] + c0; };
                N.swuflx = QN(SO_UFLX); N.swdflx = QN(SO_DFLX); N.swuflxc = QN(SO_UFLXC); N.swdflxc = QN(SO_DFLXC);
                N.nirr = QN(SO_NIRR); N.nirf = QN(SO_NIRF); N.parr = QN(SO_PARR); N.parf = QN(SO_PARF); N.uvrr = QN(SO_UVRR); N.uvrf = QN(SO_UVRF);
                N.fswband = QN(SO_FSWBAND);
                for (int k = 0; k < 8; k++) N.cot[k] = QN(SO_COT0 + k);
                span_begin(9, st);
                if (sw_reform_on()) { hipError_t e = sw_reform_reduce<R>(st, A, N); if (e != hipSuccess) return fail(GEOSRAD_EHIP, std::string("sw_reform_reduce: ") + hipGetErrorString(e)); }
                else hipLaunchKernelGGL(k_sw_reduce<R>, dim3(gx, nlay + 2), blk, 0, st, A, N);
                span_end(st);
            }
        }
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    // ---- RRTMG_SW, host pointers ---------------------------------------------------------------------------
    int sw_host(int ncol, int nlay, double scon, double adjes, int isolvar, const void *const *in, int iceflg, int liqflg, int dyofyr,
                int iaer, int cloudLM, int cloudMH, int normFlx, int32_t *clearCounts, void *const *out, int do_drfband, const void *bndscl,
                const void *indsolvar, const void *solcycfrac, void *const *dbg) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || nlay <= 0) return fail(GEOSRAD_EINVAL, "bad ncol/nlay");
        if (!dbg) {
            // production path: pinned staging + chunk pipeline (host_pipeline); the stage-dump test hooks keep the plain path below
            for (int k = 0; k < S_NIN; k++)
                if (!in[k] && !((k == S_TAUAER || k == S_SSAAER || k == S_ASMAER) && iaer != 10)) return fail(GEOSRAD_EINVAL, "null input array");
            for (int k = 0; k < SO_DRBAND; k++) if (!out[k]) return fail(GEOSRAD_EINVAL, "null output array");
            if (do_drfband && (!out[SO_DRBAND] || !out[SO_DFBAND])) return fail(GEOSRAD_EINVAL, "do_drfband set but drband/dfband null");
            const size_t L = (size_t)nlay, E = sizeof(R);
            std::vector<PipeArr> arrs;
            int ix_in[S_NIN], ix_out[SO_NOUT], ix_cc = -1;
            for (int k = 0; k < S_NIN; k++) {
                ix_in[k] = -1;
                const bool aer = (k == S_TAUAER || k == S_SSAAER || k == S_ASMAER);
                if (!in[k] || (aer && iaer != 10)) continue;
                const size_t rows = k == S_PLEV ? L + 1 : aer ? (size_t)NB_SW * L
                                  : (k == S_ALAT || k == S_COSZEN || k == S_ASDIR || k == S_ASDIF || k == S_ALDIR || k == S_ALDIF) ? 1 : L;
                ix_in[k] = (int)arrs.size(); arrs.push_back({in[k], nullptr, rows, E, 0});
            }
            for (int k = 0; k < SO_NOUT; k++) {
                ix_out[k] = -1;
                if ((k == SO_DRBAND || k == SO_DFBAND) && !do_drfband) continue;
                const size_t rows = k <= SO_DFLXC ? L + 1 : (k == SO_FSWBAND || k == SO_DRBAND || k == SO_DFBAND) ? (size_t)NB_SW : 1;
                ix_out[k] = (int)arrs.size(); arrs.push_back({nullptr, out[k], rows, E, 0});
            }
            ix_cc = (int)arrs.size(); arrs.push_back({nullptr, clearCounts, 4, sizeof(int32_t), 0});
            auto run = [&](hipStream_t st, int nc, int, char *dev, int) -> int {
                const void *din[S_NIN]; void *dout[SO_NOUT];
                for (int k = 0; k < S_NIN; k++) din[k] = ix_in[k] >= 0 ? dev + arrs[ix_in[k]].off : nullptr;
                for (int k = 0; k < SO_NOUT; k++) dout[k] = ix_out[k] >= 0 ? dev + arrs[ix_out[k]].off : nullptr;
                return sw_dev(st, nc, nlay, scon, adjes, isolvar, din, iceflg, liqflg, dyofyr, iaer, cloudLM, cloudMH, normFlx,
                              (int32_t *)(dev + arrs[ix_cc].off), dout, do_drfband, bndscl, indsolvar, solcycfrac, nullptr);
            };
            int rc = clear_slot(1);
            if (rc) return rc;
            rc = host_pipeline(ncol, arrs, run, d_err + 1);
            if (rc && rc != PIPE_FLAGGED) return rc;
            return check(stream, 1);
        }
        const size_t cl = (size_t)ncol * nlay, cv = (size_t)ncol * (nlay + 1);
        size_t insz[S_NIN];
        for (int k = 0; k < S_NIN; k++) insz[k] = cl;
        insz[S_PLEV] = cv; insz[S_ALAT] = insz[S_COSZEN] = insz[S_ASDIR] = insz[S_ASDIF] = insz[S_ALDIR] = insz[S_ALDIF] = ncol;
        insz[S_TAUAER] = insz[S_SSAAER] = insz[S_ASMAER] = cl * NB_SW;
        size_t outsz[SO_NOUT];
        for (int k = 0; k < SO_NOUT; k++) outsz[k] = ncol;
        outsz[SO_UFLX] = outsz[SO_DFLX] = outsz[SO_UFLXC] = outsz[SO_DFLXC] = cv;
        outsz[SO_FSWBAND] = outsz[SO_DRBAND] = outsz[SO_DFBAND] = (size_t)ncol * NB_SW;
        size_t off = 0;
        auto take = [&](size_t nreal) { size_t o = off; off += al(nreal * sizeof(R)); return o; };
        size_t ino[S_NIN], outo[SO_NOUT];
        for (int k = 0; k < S_NIN; k++) {
            const bool aer = (k == S_TAUAER || k == S_SSAAER || k == S_ASMAER);
            ino[k] = (in[k] && !(aer && iaer != 10)) ? take(insz[k]) : (size_t)-1;
        }
        for (int k = 0; k < SO_NOUT; k++) outo[k] = take(outsz[k]);
        const size_t cco = take((size_t)ncol * 4 * sizeof(int32_t) / sizeof(R) + 4);
        size_t dbgo[6] = {0, 0, 0, 0, 0, 0};
        if (dbg) { dbgo[0] = take(cl * NG_SW); dbgo[1] = take(cl * NG_SW); dbgo[2] = take((size_t)ncol * NG_SW); }
        if (dbg && dbg[3]) { dbgo[3] = take(cl * NG_SW); dbgo[4] = take(cl * NG_SW); dbgo[5] = take(cl * NG_SW); }
        int rc = ensure_io(off);
        if (rc) return rc;
        const void *din[S_NIN]; void *dout[SO_NOUT];
        for (int k = 0; k < S_NIN; k++) {
            din[k] = ino[k] != (size_t)-1 ? d_io + ino[k] : nullptr;
            if (din[k]) HIPCHK(hipMemcpyAsync(d_io + ino[k], in[k], insz[k] * sizeof(R), hipMemcpyHostToDevice, stream));
        }
        for (int k = 0; k < SO_NOUT; k++) dout[k] = (out[k] || k < SO_DRBAND) ? d_io + outo[k] : nullptr;
        void *ddbg[6] = {d_io + dbgo[0], d_io + dbgo[1], d_io + dbgo[2], nullptr, nullptr, nullptr};
        if (dbg && dbg[3]) for (int k = 3; k < 6; k++) ddbg[k] = d_io + dbgo[k];
        rc = sw_dev(stream, ncol, nlay, scon, adjes, isolvar, din, iceflg, liqflg, dyofyr, iaer, cloudLM, cloudMH, normFlx,
                    (int32_t *)(d_io + cco), dout, do_drfband, bndscl, indsolvar, solcycfrac, dbg ? ddbg : nullptr);
        if (rc) return rc;
        rc = check(stream, 1);
        if (rc) return rc;
        for (int k = 0; k < SO_NOUT; k++) {
            if (!out[k]) continue;
            if ((k == SO_DRBAND || k == SO_DFBAND) && !do_drfband) continue;
            HIPCHK(hipMemcpyAsync(out[k], dout[k], outsz[k] * sizeof(R), hipMemcpyDeviceToHost, stream));
        }
        if (clearCounts) HIPCHK(hipMemcpyAsync(clearCounts, d_io + cco, (size_t)ncol * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        if (dbg) {
            HIPCHK(hipMemcpyAsync(dbg[0], ddbg[0], cl * NG_SW * sizeof(R), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(dbg[1], ddbg[1], cl * NG_SW * sizeof(R), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(dbg[2], ddbg[2], (size_t)ncol * NG_SW * sizeof(R), hipMemcpyDeviceToHost, stream));
            if (dbg[3])
                for (int k = 3; k < 6; k++) HIPCHK(hipMemcpyAsync(dbg[k], ddbg[k], cl * NG_SW * sizeof(R), hipMemcpyDeviceToHost, stream));
        }
        HIPCHK(hipStreamSynchronize(stream));
        return GEOSRAD_OK;
    }

    // ---- stand-alone McICA generator, host pointers ------------------------------------------------------------------

    // =====================================================================================================
    // Chou-Suarez longwave (irrad)
    // =====================================================================================================
    int set_tables_chou_lw(const void *blob, size_t nbytes) override
    {
        HIPCHK(hipSetDevice(device));
        Blob B;
        if (!B.parse(blob, nbytes)) return fail(GEOSRAD_ETABLE, B.err);
        if (B.realbytes != (int)sizeof(R)) return fail(GEOSRAD_ETABLE, "table blob real size does not match the context's real_kind");
        TableStage<R> S(B);
        ChouDev<R> &T = h_C;
        memset(&T, 0, sizeof(T));
        auto cp = [&](R *dst, const char *nm, size_t n) { const R *s = S.get(nm, n); if (s) memcpy(dst, s, n * sizeof(R)); };
        cp(T.xkw, "xkw", 9); cp(T.xke, "xke", 9); cp(T.aw, "aw", 9); cp(T.bw, "bw", 9); cp(T.pm, "pm", 9);
        cp(T.fkw, "fkw", 54); cp(T.gkw, "gkw", 18); cp(T.cb, "cb", 60); cp(T.dcb, "dcb", 50);
        cp(T.aib, "aib_ir", 30); cp(T.awb, "awb_ir", 40); cp(T.aiw, "aiw_ir", 40); cp(T.aww, "aww_ir", 40); cp(T.aig, "aig_ir", 40);
        cp(T.awg, "awg_ir", 40);
        { int32_t mw[9]; if (S.ints("mw", 9, mw)) for (int k = 0; k < 9; k++) T.mw[k] = mw[k]; }
        T.w11 = S.scalar("w11"); T.w12 = S.scalar("w12"); T.w13 = S.scalar("w13"); T.p11 = S.scalar("p11"); T.p12 = S.scalar("p12");
        T.p13 = S.scalar("p13"); T.dwe = S.scalar("dwe"); T.dpe = S.scalar("dpe");
        S.raw(&T.c1, "c1", 26 * 30); S.raw(&T.c2, "c2", 26 * 30); S.raw(&T.c3, "c3", 26 * 30);
        S.raw(&T.oo1, "oo1", 26 * 21); S.raw(&T.oo2, "oo2", 26 * 21); S.raw(&T.oo3, "oo3", 26 * 21);
        S.raw(&T.h11, "h11", 26 * 31); S.raw(&T.h12, "h12", 26 * 31); S.raw(&T.h13, "h13", 26 * 31);
        S.raw(&T.h21, "h21", 26 * 31); S.raw(&T.h22, "h22", 26 * 31); S.raw(&T.h23, "h23", 26 * 31);
        S.raw(&T.h81, "h81", 26 * 31); S.raw(&T.h82, "h82", 26 * 31); S.raw(&T.h83, "h83", 26 * 31);
        if (!S.missing.empty()) return fail(GEOSRAD_ETABLE, "missing/ill-shaped table entries: " + S.missing);
        if (d_tab_ch) { HIPCHK(hipFree(d_tab_ch)); d_tab_ch = nullptr; }
        tab_ch_bytes = S.stage.size();
        HIPCHK(hipMalloc((void **)&d_tab_ch, tab_ch_bytes));
        HIPCHK(hipMemcpy(d_tab_ch, S.stage.data(), tab_ch_bytes, hipMemcpyHostToDevice));
        for (auto &f : S.fix) *f.first = (const R *)(d_tab_ch + f.second);
        HIPCHK(hipMemcpy(d_C, &h_C, sizeof(ChouDev<R>), hipMemcpyHostToDevice));
        have_chou = true;
        return GEOSRAD_OK;
    }

    int irrad_dev(hipStream_t st, int m, int np, const void *const *in, double co2, int trace, int ict, int icb, int ns, int na, int nb,
                  void *const *aer, void *const *out) override
    {
        HIPCHK(hipSetDevice(device));
        if (!have_chou) return fail(GEOSRAD_EINVAL, "Chou-Suarez LW tables not set: call geosrad_load_tables_chou_lw first");
        if (m <= 0 || np < 4 || np > 400) return fail(GEOSRAD_EINVAL, "bad m/np");
        if (ns < 1 || ns > 15) return fail(GEOSRAD_EINVAL, "ns must be in 1..15");
        if (!(ict >= 1 && ict < icb && icb <= np)) return fail(GEOSRAD_EINPUT, "ict / icb must satisfy 1 <= ict < icb <= np");
        if (nb < 10) return fail(GEOSRAD_EINVAL, "nb (bands of the aerosol arrays) must be 10");
        for (int k = 0; k < C_NIN; k++) if (!in[k]) return fail(GEOSRAD_EINVAL, "null input array");
        for (int k = 0; k < CO_NOUT; k++) if (!out[k]) return fail(GEOSRAD_EINVAL, "null output array");
        if (na > 0 && (!aer[0] || !aer[1] || !aer[2])) return fail(GEOSRAD_EINVAL, "na > 0 but taua/ssaa/asya null");
        const int K1 = np + 1, K2 = np + 2;
        const int nc_max = m < chunk ? m : chunk;
        const size_t need = al((size_t)nc_max * CF_NFIELD * K1 * sizeof(R)) + al((size_t)nc_max * CH_NB * CH_NKIND * K2 * sizeof(R));
        if (need > ws_ch_bytes) {
            if (d_ws_ch) { HIPCHK(hipFree(d_ws_ch)); d_ws_ch = nullptr; ws_ch_bytes = 0; }
            if (hipMalloc((void **)&d_ws_ch, need) != hipSuccess) return fail(GEOSRAD_ENOMEM, "hipMalloc of the irrad workspace failed");
            ws_ch_bytes = need;
        }
        const int nband = trace ? 10 : 9;      // irrad.F90:478 (band 10 only with trace gases)
#ifndef GEOSRAD_EXP_LDS_PAD
#define GEOSRAD_EXP_LDS_PAD 0
#endif
        const size_t lds = chou_bands_lds_bytes<R>(np) + GEOSRAD_EXP_LDS_PAD;
        if (lds > 64 * 1024) {
            if (hipFuncSetAttribute((const void *)k_chou_bands<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return fail(GEOSRAD_EINVAL, "np too large for the LDS-resident band state");
        }
        for (int c0 = 0; c0 < m; c0 += nc_max) {
            const int nc = (m - c0) < nc_max ? (m - c0) : nc_max;
            ChouArgs<R> A{};
            A.m = nc; A.ld = m; A.np = np; A.trace = trace; A.ict = ict; A.icb = icb; A.ns = ns; A.na = na; A.nb = nb; A.co2 = (R)co2;
            auto P = [&](int k) { return (const R *)in[k] + c0; };
            A.ple = P(C_PLE); A.ta = P(C_TA); A.wa = P(C_WA); A.oa = P(C_OA); A.tb = P(C_TB); A.n2o = P(C_N2O); A.ch4 = P(C_CH4);
            A.cfc11 = P(C_CFC11); A.cfc12 = P(C_CFC12); A.cfc22 = P(C_CFC22); A.cwc = P(C_CWC); A.fcld = P(C_FCLD); A.reff = P(C_REFF);
            A.fs = P(C_FS); A.tg = P(C_TG); A.eg = P(C_EG); A.tv = P(C_TV); A.ev = P(C_EV); A.rv = P(C_RV);
            A.taua = aer[0] ? (R *)aer[0] + c0 : nullptr; A.ssaa = aer[1] ? (R *)aer[1] + c0 : nullptr; A.asya = aer[2] ? (R *)aer[2] + c0 : nullptr;
            A.taudiag = (R *)out[CO_TAUDIAG] + c0;
            A.rec = (R *)d_ws_ch; A.part = (R *)(d_ws_ch + al((size_t)nc_max * CF_NFIELD * K1 * sizeof(R)));
            A.err = d_err + 2;
            span_begin(10, st);
            hipLaunchKernelGGL(k_chou_prep<R>, dim3((unsigned)((nc + 63) / 64), (unsigned)chou_prep_tiles<R>(np)), dim3(256), 0, st, A);
            span_end(st);
            span_begin(11, st);
            hipLaunchKernelGGL(k_chou_bands<R>, dim3((unsigned)((nc + CH_CPW - 1) / CH_CPW), nband), dim3(64), lds, st, A, (const ChouDev<R> *)d_C);
            span_end(st);
            ChouOut<R> O{};
            auto Q = [&](int k) { return (R *)out[k] + c0; };
            O.flxu = Q(CO_FLXU); O.flcu = Q(CO_FLCU); O.flau = Q(CO_FLAU); O.flxau = Q(CO_FLXAU); O.flxd = Q(CO_FLXD); O.flcd = Q(CO_FLCD);
            O.flad = Q(CO_FLAD); O.flxad = Q(CO_FLXAD); O.dfdts = Q(CO_DFDTS); O.sfcem = Q(CO_SFCEM);
            hipLaunchKernelGGL(k_chou_reduce<R>, dim3((unsigned)((nc + 63) / 64), 9), dim3(256), 0, st, A, O, nband);
            if (!trace)      // band 10 of taudiag stays zero (the reference zeroes the array and never reaches band 10)
                for (int k = 0; k < np; k++)
                    HIPCHK(hipMemsetAsync((R *)out[CO_TAUDIAG] + ((size_t)9 * np + k) * m + c0, 0, (size_t)nc * sizeof(R), st));
        }
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int irrad_host(int m, int np, const void *const *in, double co2, int trace, int ict, int icb, int ns, int na, int nb, void *const *aer,
                   void *const *out) override
    {
        // host arrays: pinned staging + chunk pipeline (host_pipeline), like rrtmg_lw / rrtmg_sw - transfers of a chunk overlap the
        // kernels of its neighbours (irrad is 0.33 ms of kernels per 1 000 columns against 0.08 ms of transfers)
        HIPCHK(hipSetDevice(device));
        if (m <= 0 || np <= 0 || ns < 1 || nb < 1) return fail(GEOSRAD_EINVAL, "bad m/np/ns/nb");
        for (int k = 0; k < C_NIN; k++) if (!in[k]) return fail(GEOSRAD_EINVAL, "null input array");
        for (int k = 0; k < CO_NOUT; k++) if (!out[k]) return fail(GEOSRAD_EINVAL, "null output array");
        const size_t L = (size_t)np, E = sizeof(R);
        std::vector<PipeArr> arrs;
        int ix_in[C_NIN], ix_aer[3] = {-1, -1, -1}, ix_out[CO_NOUT];
        for (int k = 0; k < C_NIN; k++) {
            const size_t rows = k == C_PLE ? L + 1 : k == C_TB ? 1 : (k == C_CWC || k == C_REFF) ? 4 * L
                              : (k == C_FS || k == C_TG || k == C_TV) ? (size_t)ns : (k == C_EG || k == C_EV || k == C_RV) ? (size_t)ns * 10 : L;
            ix_in[k] = (int)arrs.size(); arrs.push_back({in[k], nullptr, rows, E, 0});
        }
        for (int k = 0; k < 3; k++)
            if (na > 0 && aer[k]) { ix_aer[k] = (int)arrs.size(); arrs.push_back({aer[k], aer[k], L * (size_t)nb, E, 0}); }      // rescaled in place
        for (int k = 0; k < CO_NOUT; k++) {
            const size_t rows = k == CO_SFCEM ? 1 : k == CO_TAUDIAG ? 10 * L : L + 1;
            ix_out[k] = (int)arrs.size(); arrs.push_back({nullptr, out[k], rows, E, 0});
        }
        auto run = [&](hipStream_t st, int nc, int, char *dev, int) -> int {
            const void *din[C_NIN]; void *dout[CO_NOUT]; void *daer[3];
            for (int k = 0; k < C_NIN; k++) din[k] = dev + arrs[ix_in[k]].off;
            for (int k = 0; k < CO_NOUT; k++) dout[k] = dev + arrs[ix_out[k]].off;
            for (int k = 0; k < 3; k++) daer[k] = ix_aer[k] >= 0 ? dev + arrs[ix_aer[k]].off : nullptr;
            return irrad_dev(st, nc, np, din, co2, trace, ict, icb, ns, na, nb, daer, dout);
        };
        int rc = host_pipeline(m, arrs, run);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(stream));      // the Chou schemes have no input assertions (neither has the reference)
        return GEOSRAD_OK;
    }


    // =====================================================================================================
    // Chou-Suarez shortwave (sorad)
    // =====================================================================================================
    int set_tables_chou_sw(const void *blob, size_t nbytes) override
    {
        HIPCHK(hipSetDevice(device));
        Blob B;
        if (!B.parse(blob, nbytes)) return fail(GEOSRAD_ETABLE, B.err);
        if (B.realbytes != (int)sizeof(R)) return fail(GEOSRAD_ETABLE, "table blob real size does not match the context's real_kind");
        TableStage<R> S(B);
        SoradDev<R> &T = h_O;
        memset(&T, 0, sizeof(T));
        auto cp = [&](R *dst, const char *nm, size_t n) { const R *s = S.get(nm, n); if (s) memcpy(dst, s, n * sizeof(R)); };
        cp(T.zk_uv, "zk_uv", 5); cp(T.wk_uv, "wk_uv", 5); cp(T.ry_uv, "ry_uv", 5); cp(T.xk_ir, "xk_ir", 10); cp(T.ry_ir, "ry_ir", 3);
        cp(T.aig_uv, "aig_uv", 3); cp(T.awg_uv, "awg_uv", 3); cp(T.arg_uv, "arg_uv", 3); cp(T.awb_uv, "awb_uv", 2); cp(T.arb_uv, "arb_uv", 2);
        cp(T.awb_nir, "awb_nir", 6); cp(T.arb_nir, "arb_nir", 6); cp(T.aia_nir, "aia_nir", 9); cp(T.awa_nir, "awa_nir", 9);
        cp(T.ara_nir, "ara_nir", 9); cp(T.aig_nir, "aig_nir", 9); cp(T.awg_nir, "awg_nir", 9); cp(T.arg_nir, "arg_nir", 9);
        T.aib_uv = S.scalar("aib_uv"); T.aib_nir = S.scalar("aib_nir");
        S.raw(&T.coa, "coa", 62 * 101); S.raw(&T.cah, "cah", 43 * 37); S.raw(&T.caib, "caib", 11 * 9 * 11); S.raw(&T.caif, "caif", 9 * 11);
        if (!S.missing.empty()) return fail(GEOSRAD_ETABLE, "missing/ill-shaped table entries: " + S.missing);
        if (d_tab_so) { HIPCHK(hipFree(d_tab_so)); d_tab_so = nullptr; }
        tab_so_bytes = S.stage.size();
        HIPCHK(hipMalloc((void **)&d_tab_so, tab_so_bytes));
        HIPCHK(hipMemcpy(d_tab_so, S.stage.data(), tab_so_bytes, hipMemcpyHostToDevice));
        for (auto &f : S.fix) *f.first = (const R *)(d_tab_so + f.second);
        HIPCHK(hipMemcpy(d_O, &h_O, sizeof(SoradDev<R>), hipMemcpyHostToDevice));
        have_sorad = true;
        return GEOSRAD_OK;
    }

    int sorad_dev(hipStream_t st, int m, int np, int nb, const void *const *in, double co2, int ict, int icb, const void *hk_uv,
                  const void *hk_ir, void *const *out, int do_drfband) override
    {
        HIPCHK(hipSetDevice(device));
        if (!have_sorad) return fail(GEOSRAD_EINVAL, "Chou-Suarez SW tables not set: call geosrad_load_tables_chou_sw first");
        if (m <= 0 || np < 4 || np > 400) return fail(GEOSRAD_EINVAL, "bad m/np");
        if (nb < 8) return fail(GEOSRAD_EINVAL, "nb (bands of the aerosol arrays) must be 8");
        if (!(ict >= 1 && ict < icb && icb <= np)) return fail(GEOSRAD_EINPUT, "ict / icb must satisfy 1 <= ict < icb < np + 1");
        if (!hk_uv || !hk_ir) return fail(GEOSRAD_EINVAL, "hk_uv / hk_ir null");
        for (int k = 0; k < SI_NIN; k++) if (!in[k]) return fail(GEOSRAD_EINVAL, "null input array");
        for (int k = 0; k < SOO_DRBAND; k++) if (!out[k]) return fail(GEOSRAD_EINVAL, "null output array");
        if (do_drfband && (!out[SOO_DRBAND] || !out[SOO_DFBAND])) return fail(GEOSRAD_EINVAL, "do_drfband set but drband/dfband null");
        const int K2 = np + 2;
        const int nc_max = m < chunk ? m : chunk;
        const size_t per = (size_t)nc_max * sizeof(R);
        // passes: k_sorad_pass (lane = column, the per-level arrays of every pass in HBM scratch planes, 30 x K2 reals per (column, pass))
        // or k_sorad_col (one block per column, everything on chip, no scratch; GEOSRAD_SORAD_PATH=col)
        // (a layer count whose on-chip arrays exceed the LDS takes the scratch-plane path)
        const bool col_path = sorad_col_path && K2 <= 256 && sorad_col_lds_reals<R>(np) * sizeof(R) <= (size_t)160 * 1024;
        const size_t o_lay = 0, o_swh = o_lay + al(4 * K2 * per), o_colv = o_swh + al(K2 * per), o_cld = o_colv + al(8 * per),
                     o_psum = o_cld + al((size_t)SO_NGRP * 4 * K2 * per), o_aer = o_psum + al((size_t)SO_NPASS * 3 * per),
                     o_perm = o_aer + (col_path ? 0 : al((size_t)3 * SO_NGATHER * np * per)), o_cls = o_perm + al((size_t)nc_max * sizeof(int32_t)),
                     o_off = o_cls + al((size_t)nc_max), o_scr = o_off + al(16 * sizeof(int32_t)),
                     need = o_scr + (col_path ? 0 : al((size_t)SO_NPASS * SO_NPLANE * K2 * per));
        const size_t so_lds = sorad_col_lds_reals<R>(np) * sizeof(R);
        if (col_path) {
            if (so_lds > so_lds_set) {
                HIPCHK(hipFuncSetAttribute((const void *)k_sorad_col<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)so_lds));
                so_lds_set = so_lds;
            }
        }
        if (need > ws_so_bytes) {
            if (d_ws_so) { HIPCHK(hipFree(d_ws_so)); d_ws_so = nullptr; ws_so_bytes = 0; }
            if (hipMalloc((void **)&d_ws_so, need) != hipSuccess)
                return fail(GEOSRAD_ENOMEM, "hipMalloc of the sorad workspace failed (" + std::to_string(need >> 20) + " MiB); lower it with geosrad_set_chunk()");
            ws_so_bytes = need;
        }
        for (int c0 = 0; c0 < m; c0 += nc_max) {
            const int nc = (m - c0) < nc_max ? (m - c0) : nc_max;
            SoradArgs<R> A{};
            A.m = nc; A.ld = m; A.np = np; A.ict = ict; A.icb = icb; A.do_drfband = do_drfband; A.co2 = (R)co2;
            for (int p = 0; p < 5; p++) A.hk[p] = ((const R *)hk_uv)[p];
            for (int ib = 0; ib < 3; ib++) for (int ik = 0; ik < 10; ik++) A.hk[5 + ib * 10 + ik] = ((const R *)hk_ir)[ik * 3 + ib];   // hk_ir(3,10)
            auto P = [&](int k) { return (const R *)in[k] + c0; };
            A.cosz = P(SI_COSZ); A.pl = P(SI_PL); A.ta = P(SI_TA); A.wa = P(SI_WA); A.oa = P(SI_OA); A.cwc = P(SI_CWC); A.fcld = P(SI_FCLD);
            A.reff = P(SI_REFF); A.taua = P(SI_TAUA); A.ssaa = P(SI_SSAA); A.asya = P(SI_ASYA); A.rsuvbm = P(SI_RSUVBM);
            A.rsuvdf = P(SI_RSUVDF); A.rsirbm = P(SI_RSIRBM); A.rsirdf = P(SI_RSIRDF);
            A.lay = (R *)(d_ws_so + o_lay); A.swh = (R *)(d_ws_so + o_swh); A.colv = (R *)(d_ws_so + o_colv); A.cld = (R *)(d_ws_so + o_cld);
            A.psum = (R *)(d_ws_so + o_psum); A.scr = col_path ? nullptr : (R *)(d_ws_so + o_scr);
            A.aer = col_path ? nullptr : (R *)(d_ws_so + o_aer); A.perm = (int32_t *)(d_ws_so + o_perm); A.cls = (uint8_t *)(d_ws_so + o_cls);
            A.cls_off = (int32_t *)(d_ws_so + o_off);
            const dim3 blk(256);
            const unsigned gx = (unsigned)((nc + 255) / 256);
            span_begin(12, st);
            // the columns sorted by which cloud groups hold cloud (8 classes): workspace by position from here on (the on-chip path keeps the
            // caller's order: one class)
            hipLaunchKernelGGL(k_sorad_class<R>, dim3(gx), blk, 0, st, A, col_path ? 1 : 0);
            hipLaunchKernelGGL(k_partition8, dim3(1), dim3(1024), 0, st, nc, (const uint8_t *)A.cls, A.perm, A.cls_off);
            if (!col_path) hipLaunchKernelGGL(k_sorad_gather<R>, dim3(gx, (unsigned)(3 * SO_NGATHER * np)), blk, 0, st, A);
            hipLaunchKernelGGL(k_sorad_prep<R>, dim3(gx), blk, 0, st, A);
            hipLaunchKernelGGL(k_sorad_cloud<R>, dim3(gx, (unsigned)np), blk, 0, st, A, (const SoradDev<R> *)d_O);
            span_end(st);
            SoradOut<R> O{};
            auto Q = [&](int k) { return out[k] ? (R *)out[k] + c0 : (R *)nullptr; };
            O.flx = Q(SOO_FLX); O.flc = Q(SOO_FLC); O.fdiruv = Q(SOO_FDIRUV); O.fdifuv = Q(SOO_FDIFUV); O.fdirpar = Q(SOO_FDIRPAR);
            O.fdifpar = Q(SOO_FDIFPAR); O.fdirir = Q(SOO_FDIRIR); O.fdifir = Q(SOO_FDIFIR); O.flxu = Q(SOO_FLXU); O.flcu = Q(SOO_FLCU);
            O.flx_sfc_band = Q(SOO_SFCBAND); O.drband = Q(SOO_DRBAND); O.dfband = Q(SOO_DFBAND);
            span_begin(13, st);
            if (col_path) {   // one block per column, lanes = (pass slot, level) (whole wavefronts), all 35 passes on chip
                const unsigned nthr = (unsigned)sorad_col_threads(np);
                const unsigned grid = 8u * (unsigned)((nc + 7) / 8);
                hipLaunchKernelGGL(k_sorad_col<R>, dim3(grid), dim3(nthr), so_lds, st, A, (const SoradDev<R> *)d_O, O);
            } else {
                // one instantiation per class, the classes with the most sky situations first; a block without a position of the class
                // returns at once
                // one-dimensional XCD-aware grid (band_block): the 35 passes of a position block run together on one XCD and share its layer
                // inputs from that L2 (8.7 -> 8.1 ms per 100 000 columns against a (block, pass) grid)
                const dim3 g(band_grid(nc, SO_NPASS));
                const SoradDev<R> *dO = (const SoradDev<R> *)d_O;
                hipLaunchKernelGGL((k_sorad_pass<R, 7>), g, blk, 0, st, A, dO);
                hipLaunchKernelGGL((k_sorad_pass<R, 6>), g, blk, 0, st, A, dO);
                hipLaunchKernelGGL((k_sorad_pass<R, 5>), g, blk, 0, st, A, dO);
                hipLaunchKernelGGL((k_sorad_pass<R, 3>), g, blk, 0, st, A, dO);
                hipLaunchKernelGGL((k_sorad_pass<R, 4>), g, blk, 0, st, A, dO);
                hipLaunchKernelGGL((k_sorad_pass<R, 2>), g, blk, 0, st, A, dO);
                hipLaunchKernelGGL((k_sorad_pass<R, 1>), g, blk, 0, st, A, dO);
                hipLaunchKernelGGL((k_sorad_pass<R, 0>), g, blk, 0, st, A, dO);
            }
            span_end(st);               // the slot times k_sorad_pass alone (= its average duration in a rocprofv3 kernel trace)
            if (!col_path) hipLaunchKernelGGL(k_sorad_sum<R>, dim3(gx, np + 1), blk, 0, st, A, O);
            hipLaunchKernelGGL(k_sorad_reduce<R>, dim3(gx), blk, 0, st, A, (const SoradDev<R> *)d_O, O);
        }
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int sorad_host(int m, int np, int nb, const void *const *in, double co2, int ict, int icb, const void *hk_uv, const void *hk_ir,
                   void *const *out, int do_drfband) override
    {
        // host arrays: pinned staging + chunk pipeline (host_pipeline)
        HIPCHK(hipSetDevice(device));
        if (m <= 0 || np <= 0 || nb < 1) return fail(GEOSRAD_EINVAL, "bad m/np/nb");
        for (int k = 0; k < SI_NIN; k++) if (!in[k]) return fail(GEOSRAD_EINVAL, "null input array");
        const size_t L = (size_t)np, E = sizeof(R);
        std::vector<PipeArr> arrs;
        int ix_in[SI_NIN], ix_out[SOO_NOUT];
        for (int k = 0; k < SI_NIN; k++) {
            const size_t rows = (k == SI_COSZ || k == SI_RSUVBM || k == SI_RSUVDF || k == SI_RSIRBM || k == SI_RSIRDF) ? 1 : k == SI_PL ? L + 1
                              : (k == SI_CWC || k == SI_REFF) ? 4 * L : (k == SI_TAUA || k == SI_SSAA || k == SI_ASYA) ? L * (size_t)nb : L;
            ix_in[k] = (int)arrs.size(); arrs.push_back({in[k], nullptr, rows, E, 0});
        }
        for (int k = 0; k < SOO_NOUT; k++) {
            const size_t rows = (k == SOO_FLX || k == SOO_FLC || k == SOO_FLXU || k == SOO_FLCU) ? L + 1
                              : (k == SOO_SFCBAND || k == SOO_DRBAND || k == SOO_DFBAND) ? 8 : 1;
            // an output the caller does not take (or drband / dfband without do_drfband) still has its place on the device
            void *dst = ((k == SOO_DRBAND || k == SOO_DFBAND) && !do_drfband) ? nullptr : out[k];
            ix_out[k] = (int)arrs.size(); arrs.push_back({nullptr, dst, rows, E, 0});
        }
        auto run = [&](hipStream_t st, int nc, int, char *dev, int) -> int {
            const void *din[SI_NIN]; void *dout[SOO_NOUT];
            for (int k = 0; k < SI_NIN; k++) din[k] = dev + arrs[ix_in[k]].off;
            for (int k = 0; k < SOO_NOUT; k++) dout[k] = dev + arrs[ix_out[k]].off;
            return sorad_dev(st, nc, np, nb, din, co2, ict, icb, hk_uv, hk_ir, dout, do_drfband);
        };
        int rc = host_pipeline(m, arrs, run);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(stream));      // the Chou schemes have no input assertions (neither has the reference)
        return GEOSRAD_OK;
    }

    // ---- stand-alone McICA generator ---------------------------------------------------------------------------------
    char *d_mc = nullptr; size_t mc_bytes = 0;      // alpha / rcorr scratch of the stand-alone generator
    std::map<std::tuple<int, int, int>, KissJump *> sa_jumps;      // (nsubcol, nlay, inhomogeneous?) -> jump to every sub-column
    int mcica_dev(hipStream_t st, int ncol, int nsubcol, int nlay, const void *zmid, const void *alat, int doy, const void *play,
                  const void *cldfrac, const void *ciwp, const void *clwp, double cwp_tiny, const int32_t *so, int32_t *cldy,
                  void *ciwp_s, void *clwp_s) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || nlay < 4 || nsubcol <= 0) return fail(GEOSRAD_EINVAL, "bad ncol/nlay/nsubcol");
        // seed_order validation as in the reference (cloud_subcol_gen.F90:273-295)
        int sov[4] = {1, 2, 3, 4};
        if (so) {
            for (int k = 0; k < 4; k++) {
                if (so[k] < 1) return fail(GEOSRAD_EINPUT, "seed_order element < 1");
                if (so[k] > 4) return fail(GEOSRAD_EINPUT, "seed_order element > 4");
                sov[k] = so[k];
            }
        }
        const size_t cl = (size_t)ncol * nlay;
        const size_t need = 2 * al(cl * sizeof(R));
        if (need > mc_bytes) {
            if (d_mc) { HIPCHK(hipFree(d_mc)); d_mc = nullptr; mc_bytes = 0; }
            if (hipMalloc((void **)&d_mc, need) != hipSuccess) return fail(GEOSRAD_ENOMEM, "hipMalloc of the McICA scratch failed");
            mc_bytes = need;
        }
        R *d_alpha = (R *)d_mc, *d_rcorr = (R *)(d_mc + al(cl * sizeof(R)));
        const unsigned gx = (unsigned)((ncol + 255) / 256);
        span_begin(2, st);
        hipLaunchKernelGGL(k_overlap<R>, dim3(gx, nlay), dim3(256), 0, st, ncol, ncol, nlay, doy, (const R *)zmid, (const R *)alat,
                           (const int32_t *)nullptr, (const int32_t *)nullptr, (const LwDev<R> *)d_T, d_alpha, d_rcorr, (uint8_t *)nullptr);
        span_end(st);
        McArgs<R> M{};
        M.ncol = ncol; M.ld = ncol; M.nlay = nlay; M.nsubcol = nsubcol; M.doy = doy; M.cloudLM = 1; M.cloudMH = 2;
        for (int k = 0; k < 4; k++) M.so[k] = sov[k];
        M.cwp_tiny = (R)cwp_tiny;
        M.play = (const R *)play; M.cldf = (const R *)cldfrac; M.ciwp = (const R *)ciwp; M.clwp = (const R *)clwp;
        M.alpha = d_alpha; M.rcorr = d_rcorr;
        M.cldy = cldy; M.ciwp_s = (R *)ciwp_s; M.clwp_s = (R *)clwp_s;
        McPlan MP; int nseg = 0;
        int rc = mc_plan(1, nsubcol, nlay, MP, nseg);
        if (rc) return rc;
        const size_t lds = mc_sa_lds_reals(nlay, nsubcol) * sizeof(R);
        const long nblk = ((long)ncol * nsubcol + 64 * MC_SA_K - 1) / (64 * MC_SA_K);
        span_begin(3, st);
        if (lds <= 160 * 1024 && nblk <= 0x7FFFFFFFL && !getenv("GEOSRAD_MCICA_LANE_COLUMN")) {
            // lane = (column, sub-column) pair, outputs written row-major through an LDS tile (k_mcica_sa)
            const bool inhomo = h_T.xcw != nullptr;
            const auto key = std::make_tuple(nsubcol, nlay, inhomo ? 1 : 0);
            auto it = sa_jumps.find(key);
            if (it == sa_jumps.end()) {
                const uint64_t per = (uint64_t)(inhomo ? 4 : 2) * (uint64_t)nlay;
                std::vector<KissJump> js((size_t)nsubcol);
                for (int q = 0; q < nsubcol; q++) js[q] = make_kiss_jump((uint64_t)q * per);
                KissJump *dj = nullptr;
                HIPCHK(hipMalloc((void **)&dj, js.size() * sizeof(KissJump)));
                HIPCHK(hipMemcpy(dj, js.data(), js.size() * sizeof(KissJump), hipMemcpyHostToDevice));
                it = sa_jumps.emplace(key, dj).first;
            }
            if (lds > 65536) HIPCHK(hipFuncSetAttribute((const void *)k_mcica_sa<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_mcica_sa<R>), dim3((unsigned)nblk), dim3(64), lds, st, M, (const KissJump *)it->second, MP.jhalf,
                               (const LwDev<R> *)d_T);
        } else {
            // lane = column (tiles beyond 64 KB of LDS: fp64 with more than 127 layers)
            hipLaunchKernelGGL((k_mcica<R, 1>), dim3(xcd_grid(ncol, 64, nseg)), dim3(64), 0, st, M, MP, (const LwDev<R> *)d_T, (const SwDev<R> *)nullptr);
        }
        span_end(st);
        HIPCHK(hipGetLastError());
        return GEOSRAD_OK;
    }

    int mcica_host(int ncol, int nsubcol, int nlay, const void *zmid, const void *alat, int doy, const void *play, const void *cldfrac,
                   const void *ciwp, const void *clwp, double cwp_tiny, const int32_t *so, int32_t *cldy, void *ciwp_s,
                   void *clwp_s) override
    {
        HIPCHK(hipSetDevice(device));
        if (ncol <= 0 || nlay < 4 || nsubcol <= 0) return fail(GEOSRAD_EINVAL, "bad ncol/nlay/nsubcol");
        const size_t cl = (size_t)ncol * nlay, co = cl * nsubcol;
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
        const size_t o_z = take(cl * sizeof(R)), o_p = take(cl * sizeof(R)), o_f = take(cl * sizeof(R)), o_i = take(cl * sizeof(R)),
                     o_l = take(cl * sizeof(R)), o_a = take((size_t)ncol * sizeof(R)), o_cy = take(co * 4), o_ci = take(co * sizeof(R)),
                     o_cl = take(co * sizeof(R));
        int rc = ensure_io(off);
        if (rc) return rc;
        const void *src[6] = {zmid, play, cldfrac, ciwp, clwp, alat};
        const size_t dst[6] = {o_z, o_p, o_f, o_i, o_l, o_a};
        for (int k = 0; k < 6; k++)
            HIPCHK(hipMemcpyAsync(d_io + dst[k], src[k], (k == 5 ? (size_t)ncol : cl) * sizeof(R), hipMemcpyHostToDevice, stream));
        rc = mcica_dev(stream, ncol, nsubcol, nlay, d_io + o_z, d_io + o_a, doy, d_io + o_p, d_io + o_f, d_io + o_i, d_io + o_l, cwp_tiny, so,
                       (int32_t *)(d_io + o_cy), d_io + o_ci, d_io + o_cl);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(cldy, d_io + o_cy, co * 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(ciwp_s, d_io + o_ci, co * sizeof(R), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(clwp_s, d_io + o_cl, co * sizeof(R), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return GEOSRAD_OK;
    }
};

}  // namespace

#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 4
geosrad_ctx *geosrad_new_ctx_f32() { return new Ctx<float>(); }
#endif
#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 8
geosrad_ctx *geosrad_new_ctx_f64() { return new Ctx<double>(); }
#endif
#endif   // GEOSRAD_PART != 0

#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 0
// ---------------------------------------------------------------------------------------------------
// extern "C"
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// A context over several devices (`geosrad_create(ctx**, device_ids, ndev, kind)`; 2a: "splitting that batch ... ways").
// It owns one single-device context per entry of device_ids; the host-pointer solver entry points cut [0, ncol) into ndev contiguous
// shards and run one child's pipeline per shard concurrently (one host thread each; a child reads / writes its shard of the
// caller's arrays in place: base pointer + shard start, leading dimension = the full ncol).  Columns are independent, so the
// result is bitwise the single-device one (tested with device_ids = {0, 0}).  Table / parameter setters go to every child;
// the `_dev` entry points (device pointers belong to ONE device) are refused.
// ---------------------------------------------------------------------------------------------------
namespace {
struct MultiCtx final : geosrad_ctx {
    std::vector<geosrad_ctx *> kid;
    ~MultiCtx() override { for (auto *k : kid) { (void)hipSetDevice(k->device); delete k; } }
    int init() override { return GEOSRAD_OK; }
    int all(const std::function<int(geosrad_ctx *)> &f)
    {
        for (auto *k : kid) { const int rc = f(k); if (rc) { last_error = k->last_error; return rc; } }
        return GEOSRAD_OK;
    }
    int nodev(const char *what) { return fail(GEOSRAD_EINVAL, std::string(what) + ": device-pointer entry points need a single-device context (geosrad_create)"); }
    // shard s of n columns: [start, start + count)
    static void shard(int n, int nk, int s, int &start, int &count) { const int per = (n + nk - 1) / nk; start = s * per; count = start >= n ? 0 : (n - start < per ? n - start : per); }
    // fn(child, first column, columns) per shard, concurrently; the first failing shard's status and message
    int run_shards(int ncol, const std::function<int(geosrad_ctx *, int, int)> &fn)
    {
        const int nk = (int)kid.size();
        std::vector<int> rc(nk, GEOSRAD_OK);
        std::vector<std::thread> th;
        for (int s = 0; s < nk; s++) {
            int c0, nc; shard(ncol, nk, s, c0, nc);
            if (nc <= 0) continue;
            th.emplace_back([&, s, c0, nc] { kid[s]->host_ld = (size_t)ncol; rc[s] = fn(kid[s], c0, nc); kid[s]->host_ld = 0; });
        }
        for (auto &t : th) t.join();
        for (int s = 0; s < nk; s++) if (rc[s]) { last_error = kid[s]->last_error; return rc[s]; }
        return GEOSRAD_OK;
    }
    static const void *off(const void *p, size_t bytes) { return p ? (const char *)p + bytes : nullptr; }
    static void *off(void *p, size_t bytes) { return p ? (char *)p + bytes : nullptr; }

    int set_tables_lw(const void *b, size_t n) override { return all([&](geosrad_ctx *k) { return k->set_tables_lw(b, n); }); }
    int set_tables_sw(const void *b, size_t n) override { return all([&](geosrad_ctx *k) { return k->set_tables_sw(b, n); }); }
    int set_tables_chou_lw(const void *b, size_t n) override { return all([&](geosrad_ctx *k) { return k->set_tables_chou_lw(b, n); }); }
    int set_tables_chou_sw(const void *b, size_t n) override { return all([&](geosrad_ctx *k) { return k->set_tables_chou_sw(b, n); }); }
    int set_inhomogeneity(int ih, const void *b, size_t n) override { return all([&](geosrad_ctx *k) { return k->set_inhomogeneity(ih, b, n); }); }
    int set_corr(const double *a, const double *r) override { return all([&](geosrad_ctx *k) { return k->set_corr(a, r); }); }
    size_t workspace_bytes() const override { size_t t = 0; for (auto *k : kid) t += k->workspace_bytes(); return t; }
    int check(hipStream_t, int which) override { return all([&](geosrad_ctx *k) { (void)hipSetDevice(k->device); return k->check(k->stream, which); }); }

    int lw_host(int ncol, int nlay, int dudTs, const void *const *in, int iceflg, int liqflg, int dyofyr, int cloudLM, int cloudMH,
                int32_t *clearCounts, void *const *out, const int32_t *band_output, void *taug, void *pfracs) override
    {
        if (taug || pfracs) return fail(GEOSRAD_EINVAL, "stage dumps need a single-device context");
        const size_t E = (size_t)real_kind;
        return run_shards(ncol, [&](geosrad_ctx *k, int c0, int nc) {
            const void *i2[I_NIN]; void *o2[O_NOUT];
            for (int j = 0; j < I_NIN; j++) i2[j] = off(in[j], (size_t)c0 * E);
            for (int j = 0; j < O_NOUT; j++) o2[j] = off(out[j], (size_t)c0 * E * ((j == O_OLRB || j == O_DOLRB) ? 16 : 1));
            return k->lw_host(nc, nlay, dudTs, i2, iceflg, liqflg, dyofyr, cloudLM, cloudMH, clearCounts ? clearCounts + c0 : nullptr, o2,
                              band_output, nullptr, nullptr);
        });
    }
    int sw_host(int ncol, int nlay, double scon, double adjes, int isolvar, const void *const *in, int iceflg, int liqflg, int dyofyr,
                int iaer, int cloudLM, int cloudMH, int normFlx, int32_t *clearCounts, void *const *out, int do_drfband, const void *bndscl,
                const void *indsolvar, const void *solcycfrac, void *const *dbg) override
    {
        if (dbg) return fail(GEOSRAD_EINVAL, "stage dumps need a single-device context");
        const size_t E = (size_t)real_kind;
        return run_shards(ncol, [&](geosrad_ctx *k, int c0, int nc) {
            const void *i2[S_NIN]; void *o2[SO_NOUT];
            for (int j = 0; j < S_NIN; j++) i2[j] = off(in[j], (size_t)c0 * E);
            for (int j = 0; j < SO_NOUT; j++) o2[j] = off(out[j], (size_t)c0 * E);
            return k->sw_host(nc, nlay, scon, adjes, isolvar, i2, iceflg, liqflg, dyofyr, iaer, cloudLM, cloudMH, normFlx,
                              clearCounts ? clearCounts + c0 : nullptr, o2, do_drfband, bndscl, indsolvar, solcycfrac, nullptr);
        });
    }
    int irrad_host(int m, int np, const void *const *in, double co2, int trace, int ict, int icb, int ns, int na, int nb, void *const *aer,
                   void *const *out) override
    {
        const size_t E = (size_t)real_kind;
        return run_shards(m, [&](geosrad_ctx *k, int c0, int nc) {
            const void *i2[C_NIN]; void *a2[3], *o2[CO_NOUT];
            for (int j = 0; j < C_NIN; j++) i2[j] = off(in[j], (size_t)c0 * E);
            for (int j = 0; j < 3; j++) a2[j] = off(aer[j], (size_t)c0 * E);
            for (int j = 0; j < CO_NOUT; j++) o2[j] = off(out[j], (size_t)c0 * E);
            return k->irrad_host(nc, np, i2, co2, trace, ict, icb, ns, na, nb, a2, o2);
        });
    }
    int sorad_host(int m, int np, int nb, const void *const *in, double co2, int ict, int icb, const void *hk_uv, const void *hk_ir,
                   void *const *out, int do_drfband) override
    {
        const size_t E = (size_t)real_kind;
        return run_shards(m, [&](geosrad_ctx *k, int c0, int nc) {
            const void *i2[SI_NIN]; void *o2[SOO_NOUT];
            for (int j = 0; j < SI_NIN; j++) i2[j] = off(in[j], (size_t)c0 * E);
            for (int j = 0; j < SOO_NOUT; j++) o2[j] = off(out[j], (size_t)c0 * E);
            return k->sorad_host(nc, np, nb, i2, co2, ict, icb, hk_uv, hk_ir, o2, do_drfband);
        });
    }
    int mcica_host(int, int, int, const void *, const void *, int, const void *, const void *, const void *, const void *, double,
                   const int32_t *, int32_t *, void *, void *) override { return fail(GEOSRAD_EINVAL, "geosrad_mcica needs a single-device context"); }
    int lw_dev(hipStream_t, int, int, int, const void *const *, int, int, int, int, int, int32_t *, void *const *, const int32_t *, void *, void *,
               const LwRats *) override { return nodev("geosrad_rrtmg_lw_dev"); }
    int mcica_dev(hipStream_t, int, int, int, const void *, const void *, int, const void *, const void *, const void *, const void *, double,
                  const int32_t *, int32_t *, void *, void *) override { return nodev("geosrad_mcica_dev"); }
    int sorad_dev(hipStream_t, int, int, int, const void *const *, double, int, int, const void *, const void *, void *const *, int) override { return nodev("geosrad_sorad_dev"); }
    int irrad_dev(hipStream_t, int, int, const void *const *, double, int, int, int, int, int, int, void *const *, void *const *) override { return nodev("geosrad_irrad_dev"); }
    int sw_dev(hipStream_t, int, int, double, double, int, const void *const *, int, int, int, int, int, int, int, int32_t *, void *const *, int,
               const void *, const void *, const void *, void *const *) override { return nodev("geosrad_rrtmg_sw_dev"); }
    int lw_driver_dev(hipStream_t, int, int, int, const void *const *, const double *, int, int, int, int, int, const int32_t *, void *const *, int,
                      const int32_t *, void *const *) override { return nodev("geosrad_lw_driver_rrtmg_dev"); }
    int sw_driver_dev(hipStream_t, int, int, int, const void *const *, const double *, int, int, double, double, int, int, int, int, int, int,
                      const void *, const void *, void *const *) override { return nodev("geosrad_sw_driver_rrtmg_dev"); }
    int lw_chou_post_dev(hipStream_t, int, int, const void *const *, void *const *) override { return nodev("geosrad_lw_chou_post_dev"); }
    int sw_driver_chou_dev(hipStream_t, int, int, const void *const *, const double *, int, int, const void *, const void *, int,
                           void *const *) override { return nodev("geosrad_sw_driver_chou_dev"); }
    int lw_update_flx_dev(hipStream_t, int, int, int, int, int, double, const void *const *, void *const *) override { return nodev("geosrad_lw_update_flx_dev"); }
    int lw_update_rats_dev(hipStream_t, int, int, int, const void *const *, void *const *) override { return nodev("geosrad_lw_update_rats_dev"); }
    int lw_update_bands_dev(hipStream_t, int, const int32_t *, const double *, const double *, double, const void *, const void *, const void *,
                            const void *, void *, void *) override { return nodev("geosrad_lw_update_bands_dev"); }
    int sw_update_export_dev(hipStream_t, int, int, int, const void *const *, void *const *) override { return nodev("geosrad_sw_update_export_dev"); }
    int sw_update_surface_dev(hipStream_t, int, int, double, const void *const *, void *const *) override { return nodev("geosrad_sw_update_surface_dev"); }
    int rad_tendencies_dev(hipStream_t, int, int, double, double, const void *const *, void *const *) override { return nodev("geosrad_rad_tendencies_dev"); }
    int lit_index_dev(hipStream_t, int, const void *, int32_t *, int32_t *, int32_t *, int *) override { return nodev("geosrad_lit_index_dev"); }
    int lit_pack_dev(hipStream_t, int, int, int, const int32_t *, const int32_t *, const void *, void *) override { return nodev("geosrad_lit_pack_dev"); }
    int lit_unpack_dev(hipStream_t, int, int, int, const int32_t *, const void *, void *, int, double) override { return nodev("geosrad_lit_unpack_dev"); }
};
}  // namespace

extern "C" {

// Which device a process should use when nothing says so explicitly: GEOSRAD_DEVICE if set, else the node-local MPI rank the launcher
// exports (Open MPI, Slurm, MVAPICH2, Intel MPI / MPICH Hydra) modulo the number of visible devices - 96 ranks of a GEOS job on an
// 8-GPU node then share the GPUs 12 to one without any configuration (the reference balances its ranks' work, not its devices:
// GEOS_SolarGridComp.F90:3701-3709).  Pure function of the environment and ndev (no HIP call): 0 when nothing is set.
int geosrad_pick_device(int ndev)
{
    if (ndev <= 0) return 0;
    // an explicit device id is taken as it is: out of range (a stale or mistyped GEOSRAD_DEVICE) is an error (-1 -> GEOSRAD_ENODEV from
    // geosrad_create), not another GPU; only the launchers' node-local ranks wrap around the device count
    if (const char *e = getenv("GEOSRAD_DEVICE")) {
        if (*e) {
            char *end = nullptr;
            const long r = strtol(e, &end, 10);
            return (end == e || *end || r < 0 || r >= ndev) ? -1 : (int)r;
        }
    }
    static const char *vars[] = {"OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "PMI_LOCAL_RANK"};
    for (const char *v : vars) {
        const char *e = getenv(v);
        if (!e || !*e) continue;
        char *end = nullptr;
        const long r = strtol(e, &end, 10);
        if (end == e || r < 0) continue;
        return (int)(r % ndev);
    }
    return 0;
}

int geosrad_create_multi(geosrad_ctx **out, const int *device_ids, int ndev, int real_kind)
{
    if (!out || !device_ids || ndev < 1 || (real_kind != 4 && real_kind != 8)) return GEOSRAD_EINVAL;
    *out = nullptr;
    MultiCtx *m = new MultiCtx();
    m->real_kind = real_kind; m->device = device_ids[0];
    for (int k = 0; k < ndev; k++) {
        geosrad_ctx *c = nullptr;
        const int rc = geosrad_create(&c, device_ids[k], real_kind);
        if (rc) { delete m; return rc; }
        m->kid.push_back(c);
    }
    *out = m;
    return GEOSRAD_OK;
}

int geosrad_create(geosrad_ctx **out, int device_id, int real_kind)
{
    if (!out || (real_kind != 4 && real_kind != 8)) return GEOSRAD_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GEOSRAD_ENODEV;
    if (device_id == GEOSRAD_DEVICE_AUTO) device_id = geosrad_pick_device(ndev);
    if (device_id < 0 || device_id >= ndev) return GEOSRAD_ENODEV;
    geosrad_ctx *c = real_kind == 4 ? geosrad_new_ctx_f32() : geosrad_new_ctx_f64();
    c->device = device_id; c->real_kind = real_kind;
    {   // A/B switch for the measurements in profiles/: GEOSRAD_LW_PATH=cols | bands
        const char *e = getenv("GEOSRAD_LW_PATH");
        if (e) { c->lw_cols_path = !strcmp(e, "cols"); c->lw_split_path = !strcmp(e, "split"); }
        e = getenv("GEOSRAD_SW_PATH");
        if (e) c->sw_path = !strcmp(e, "bands") ? 0 : 2;
        if ((e = getenv("GEOSRAD_SORAD_PATH"))) c->sorad_col_path = !strcmp(e, "col");
        // tuning of the host-pointer pipeline: columns per staged chunk, copy threads
        if ((e = getenv("GEOSRAD_HOST_CHUNK")) && atoi(e) >= 64) c->host_chunk = c->host_chunk_default = atoi(e);
        if ((e = getenv("GEOSRAD_HOST_THREADS")) && atoi(e) >= 1) c->host_threads = atoi(e) > 64 ? 64 : atoi(e);
        if ((e = getenv("GEOSRAD_HOST_NT"))) c->host_nt = atoi(e) != 0;
    }
    int rc = c->init();
    if (rc) { delete c; return rc; }
    *out = c;
    return GEOSRAD_OK;
}

int geosrad_destroy(geosrad_ctx *c) { if (!c) return GEOSRAD_EINVAL; (void)hipSetDevice(c->device); delete c; return GEOSRAD_OK; }
const char *geosrad_last_error(const geosrad_ctx *c) { return c ? c->last_error.c_str() : "null context"; }
int geosrad_real_kind(const geosrad_ctx *c) { return c ? c->real_kind : 0; }
int geosrad_set_chunk(geosrad_ctx *c, int n)
{
    if (!c || n < 64) return GEOSRAD_EINVAL;
    if (auto *m = dynamic_cast<MultiCtx *>(c)) { for (auto *k : m->kid) (void)geosrad_set_chunk(k, n); return GEOSRAD_OK; }
    c->chunk = n;
    c->host_chunk = n < c->host_chunk_default ? n : c->host_chunk_default;      // the host-pointer pipeline never stages more than a batch
    return GEOSRAD_OK;
}
size_t geosrad_workspace_bytes(const geosrad_ctx *c) { return c ? c->workspace_bytes() : 0; }

int geosrad_set_tables_lw(geosrad_ctx *c, const void *blob, size_t n) { return c ? c->set_tables_lw(blob, n) : GEOSRAD_EINVAL; }

static int read_file(geosrad_ctx *c, const char *path, std::vector<char> &buf)
{
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) return c->fail(GEOSRAD_ETABLE, std::string("cannot open table file: ") + (path ? path : "(null)"));
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != buf.size()) return c->fail(GEOSRAD_ETABLE, std::string("short read: ") + path);
    return GEOSRAD_OK;
}
int geosrad_load_tables_lw(geosrad_ctx *c, const char *path)
{
    if (!c) return GEOSRAD_EINVAL;
    std::vector<char> buf;
    int rc = read_file(c, path, buf);
    return rc ? rc : c->set_tables_lw(buf.data(), buf.size());
}
int geosrad_set_inhomogeneity(geosrad_ctx *c, int ih, const void *blob, size_t n) { return c ? c->set_inhomogeneity(ih, blob, n) : GEOSRAD_EINVAL; }
int geosrad_load_inhomogeneity(geosrad_ctx *c, int ih, const char *path)
{
    if (!c) return GEOSRAD_EINVAL;
    if (ih == 0) return c->set_inhomogeneity(0, nullptr, 0);
    std::vector<char> buf;
    int rc = read_file(c, path, buf);
    return rc ? rc : c->set_inhomogeneity(ih, buf.data(), buf.size());
}
int geosrad_set_corr_lengths(geosrad_ctx *c, const double *adl, const double *rdl) { return c ? c->set_corr(adl, rdl) : GEOSRAD_EINVAL; }

// Host-side copy of one named real array of a GRTB coefficient file (no context, no device): the Fortran shim keeps the xcw table
// on the host as well, because the reference's public zcw_lookup (cloud_condensate_inhomogeneity.F90:86-124) is a host function
// that the GridComps import for their RRTMGP branch.
int geosrad_read_table(const char *path, const char *name, int real_kind, void *dst, size_t count)
{
    if (!path || !name || !dst || (real_kind != 4 && real_kind != 8)) return GEOSRAD_EINVAL;
    FILE *f = fopen(path, "rb");
    if (!f) return GEOSRAD_ETABLE;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> buf(n > 0 ? (size_t)n : 0);
    const size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    Blob B;
    if (got != buf.size() || !B.parse(buf.data(), buf.size())) return GEOSRAD_ETABLE;
    auto it = B.e.find(name);
    if (it == B.e.end() || it->second.kind != real_kind || it->second.count != count) return GEOSRAD_ETABLE;
    memcpy(dst, it->second.data, count * (size_t)real_kind);
    return GEOSRAD_OK;
}


int geosrad_set_tables_sw(geosrad_ctx *c, const void *blob, size_t n) { return c ? c->set_tables_sw(blob, n) : GEOSRAD_EINVAL; }
int geosrad_load_tables_sw(geosrad_ctx *c, const char *path)
{
    if (!c) return GEOSRAD_EINVAL;
    std::vector<char> buf;
    int rc = read_file(c, path, buf);
    return rc ? rc : c->set_tables_sw(buf.data(), buf.size());
}

#define SW_PACK()                                                                                                              \
    const void *in[S_NIN] = {play, plev, tlay, h2ovmr, o3vmr, co2vmr, ch4vmr, o2vmr, cld, ciwp, clwp, rei, rel, zm, alat, tauaer,     \
                             ssaaer, asmaer, coszen, asdir, asdif, aldir, aldif};                                              \
    void *out[SO_NOUT] = {swuflx, swdflx, swuflxc, swdflxc, nirr, nirf, parr, parf, uvrr, uvrf, fswband, cotdtp, cotdhp, cotdmp,     \
                          cotdlp, cotntp, cotnhp, cotnmp, cotnlp, drband, dfband}

int geosrad_rrtmg_sw(geosrad_ctx *c, int rpart, int ncol, int nlay, double scon, double adjes, const void *coszen, int isolvar,
                     const void *play, const void *plev, const void *tlay, const void *h2ovmr, const void *o3vmr, const void *co2vmr,
                     const void *ch4vmr, const void *o2vmr, int iceflgsw, int liqflgsw, const void *cld, const void *ciwp,
                     const void *clwp, const void *rei, const void *rel, int dyofyr, const void *zm, const void *alat, int iaer,
                     const void *tauaer, const void *ssaaer, const void *asmaer, const void *asdir, const void *asdif,
                     const void *aldir, const void *aldif, int cloudLM, int cloudMH, int normFlx, int32_t *clearCounts, void *swuflx,
                     void *swdflx, void *swuflxc, void *swdflxc, void *nirr, void *nirf, void *parr, void *parf, void *uvrr,
                     void *uvrf, void *fswband, void *cotdtp, void *cotdhp, void *cotdmp, void *cotdlp, void *cotntp, void *cotnhp,
                     void *cotnmp, void *cotnlp, int do_drfband, void *drband, void *dfband, const void *bndscl, const void *indsolvar,
                     const void *solcycfrac)
{
    if (!c) return GEOSRAD_EINVAL;
    (void)rpart;
    SW_PACK();
    return c->sw_host(ncol, nlay, scon, adjes, isolvar, in, iceflgsw, liqflgsw, dyofyr, iaer, cloudLM, cloudMH, normFlx, clearCounts, out,
                      do_drfband, bndscl, indsolvar, solcycfrac, nullptr);
}

int geosrad_rrtmg_sw_dev(geosrad_ctx *c, void *stream, int rpart, int ncol, int nlay, double scon, double adjes, const void *coszen,
                         int isolvar, const void *play, const void *plev, const void *tlay, const void *h2ovmr, const void *o3vmr,
                         const void *co2vmr, const void *ch4vmr, const void *o2vmr, int iceflgsw, int liqflgsw, const void *cld,
                         const void *ciwp, const void *clwp, const void *rei, const void *rel, int dyofyr, const void *zm,
                         const void *alat, int iaer, const void *tauaer, const void *ssaaer, const void *asmaer, const void *asdir,
                         const void *asdif, const void *aldir, const void *aldif, int cloudLM, int cloudMH, int normFlx,
                         int32_t *clearCounts, void *swuflx, void *swdflx, void *swuflxc, void *swdflxc, void *nirr, void *nirf,
                         void *parr, void *parf, void *uvrr, void *uvrf, void *fswband, void *cotdtp, void *cotdhp, void *cotdmp,
                         void *cotdlp, void *cotntp, void *cotnhp, void *cotnmp, void *cotnlp, int do_drfband, void *drband,
                         void *dfband, const void *bndscl, const void *indsolvar,
                     const void *solcycfrac)
{
    if (!c || !clearCounts) return GEOSRAD_EINVAL;
    (void)rpart;
    SW_PACK();
    return c->sw_dev((hipStream_t)stream, ncol, nlay, scon, adjes, isolvar, in, iceflgsw, liqflgsw, dyofyr, iaer, cloudLM, cloudMH, normFlx,
                     clearCounts, out, do_drfband, bndscl, indsolvar, solcycfrac, nullptr);
}

int geosrad_rrtmg_sw_taumol(geosrad_ctx *c, int ncol, int nlay, double scon, int isolvar, const void *play, const void *plev,
                            const void *tlay, const void *h2ovmr, const void *o3vmr, const void *co2vmr, const void *ch4vmr,
                            const void *o2vmr, const void *bndscl, const void *indsolvar, const void *solcycfrac, void *taug, void *taur,
                            void *ssi)
{
    if (!c || !taug || !taur || !ssi || ncol <= 0 || nlay <= 0) return GEOSRAD_EINVAL;
    // clear-sky run (zero cloud field, overhead sun, black surface); fluxes are discarded
    const size_t esz = (size_t)c->real_kind;
    const size_t cv = (size_t)ncol * (nlay + 1) * esz;
    std::vector<char> zero(cv, 0), tens((size_t)ncol * nlay * esz, 0), one((size_t)ncol * esz, 0), scratch(4 * cv + (size_t)ncol * esz * (6 + 14 + 8));
    for (size_t i = 0; i < (size_t)ncol * nlay; i++) { if (esz == 4) ((float *)tens.data())[i] = 10.f; else ((double *)tens.data())[i] = 10.; }
    for (size_t i = 0; i < (size_t)ncol; i++) { if (esz == 4) ((float *)one.data())[i] = 1.f; else ((double *)one.data())[i] = 1.; }
    std::vector<int32_t> cc((size_t)ncol * 4);
    const void *cld = zero.data(), *ciwp = zero.data(), *clwp = zero.data(), *rei = tens.data(), *rel = tens.data(), *zm = zero.data(),
               *alat = zero.data(), *tauaer = nullptr, *ssaaer = nullptr, *asmaer = nullptr, *coszen = one.data(), *asdir = zero.data(),
               *asdif = zero.data(), *aldir = zero.data(), *aldif = zero.data();
    char *s = scratch.data();
    const size_t cn = (size_t)ncol * esz;
    void *swuflx = s, *swdflx = s + cv, *swuflxc = s + 2 * cv, *swdflxc = s + 3 * cv;
    char *q = s + 4 * cv;
    void *nirr = q, *nirf = q + cn, *parr = q + 2 * cn, *parf = q + 3 * cn, *uvrr = q + 4 * cn, *uvrf = q + 5 * cn, *fswband = q + 6 * cn;
    q += 20 * cn;
    void *cotdtp = q, *cotdhp = q + cn, *cotdmp = q + 2 * cn, *cotdlp = q + 3 * cn, *cotntp = q + 4 * cn, *cotnhp = q + 5 * cn,
         *cotnmp = q + 6 * cn, *cotnlp = q + 7 * cn, *drband = nullptr, *dfband = nullptr;
    SW_PACK();
    void *dbg[6] = {taug, taur, ssi, nullptr, nullptr, nullptr};
    return c->sw_host(ncol, nlay, scon, 1.0, isolvar, in, 3, 1, 1, 0, 1, 2, 0, cc.data(), out, 0, bndscl, indsolvar, solcycfrac, dbg);
}

int geosrad_rrtmg_sw_cldprmc(geosrad_ctx *c, int ncol, int nlay, const void *play, const void *plev, const void *tlay, const void *h2ovmr,
                             const void *o3vmr, const void *co2vmr, const void *ch4vmr, const void *o2vmr, int iceflgsw, int liqflgsw,
                             const void *cld, const void *ciwp, const void *clwp, const void *rei, const void *rel, int dyofyr,
                             const void *zm, const void *alat, int cloudLM, int cloudMH, void *taucmc, void *ssacmc, void *asmcmc)
{
    if (!c || !taucmc || !ssacmc || !asmcmc || ncol <= 0 || nlay <= 0) return GEOSRAD_EINVAL;
    // the solver's own McICA + cldprmc_sw on these columns (overhead sun, black surface, no aerosol); fluxes are discarded
    const size_t esz = (size_t)c->real_kind;
    const size_t cv = (size_t)ncol * (nlay + 1) * esz, cn = (size_t)ncol * esz, cg = (size_t)ncol * nlay * NG_SW * esz;
    std::vector<char> zero(cn, 0), one(cn, 0), scratch(4 * cv + cn * (6 + 14 + 8) + 2 * cg + cn * NG_SW);
    for (size_t i = 0; i < (size_t)ncol; i++) { if (esz == 4) ((float *)one.data())[i] = 1.f; else ((double *)one.data())[i] = 1.; }
    std::vector<int32_t> cc((size_t)ncol * 4);
    const void *tauaer = nullptr, *ssaaer = nullptr, *asmaer = nullptr, *coszen = one.data(), *asdir = zero.data(), *asdif = zero.data(),
               *aldir = zero.data(), *aldif = zero.data();
    char *s = scratch.data();
    void *swuflx = s, *swdflx = s + cv, *swuflxc = s + 2 * cv, *swdflxc = s + 3 * cv;
    char *q = s + 4 * cv;
    void *nirr = q, *nirf = q + cn, *parr = q + 2 * cn, *parf = q + 3 * cn, *uvrr = q + 4 * cn, *uvrf = q + 5 * cn, *fswband = q + 6 * cn;
    q += 20 * cn;
    void *cotdtp = q, *cotdhp = q + cn, *cotdmp = q + 2 * cn, *cotdlp = q + 3 * cn, *cotntp = q + 4 * cn, *cotnhp = q + 5 * cn,
         *cotnmp = q + 6 * cn, *cotnlp = q + 7 * cn, *drband = nullptr, *dfband = nullptr;
    q += 8 * cn;
    SW_PACK();
    void *dbg[6] = {q, q + cg, q + 2 * cg, taucmc, ssacmc, asmcmc};
    return c->sw_host(ncol, nlay, 1361.0, 1.0, 0, in, iceflgsw, liqflgsw, dyofyr, 0, cloudLM, cloudMH, 0, cc.data(), out, 0, nullptr, nullptr, nullptr, dbg);
}

int geosrad_set_tables_chou_lw(geosrad_ctx *c, const void *blob, size_t n) { return c ? c->set_tables_chou_lw(blob, n) : GEOSRAD_EINVAL; }
int geosrad_load_tables_chou_lw(geosrad_ctx *c, const char *path)
{
    if (!c) return GEOSRAD_EINVAL;
    std::vector<char> buf;
    int rc = read_file(c, path, buf);
    return rc ? rc : c->set_tables_chou_lw(buf.data(), buf.size());
}

#define CH_PACK()                                                                                                                   \
    const void *in[C_NIN] = {ple, ta, wa, oa, tb, n2o, ch4, cfc11, cfc12, cfc22, cwc, fcld, reff, fs, tg, eg, tv, ev, rv};             \
    void *aer[3] = {taua, ssaa, asya};                                                                                              \
    void *out[CO_NOUT] = {flxu, flcu, flau, flxau, flxd, flcd, flad, flxad, dfdts, sfcem, taudiag}

int geosrad_irrad(geosrad_ctx *c, int m, int np, const void *ple, const void *ta, const void *wa, const void *oa, const void *tb, double co2,
                  int trace, const void *n2o, const void *ch4, const void *cfc11, const void *cfc12, const void *cfc22, const void *cwc,
                  const void *fcld, int ict, int icb, const void *reff, int ns, const void *fs, const void *tg, const void *eg,
                  const void *tv, const void *ev, const void *rv, int na, int nb, void *taua, void *ssaa, void *asya, void *flxu,
                  void *flcu, void *flau, void *flxau, void *flxd, void *flcd, void *flad, void *flxad, void *dfdts, void *sfcem,
                  void *taudiag)
{
    if (!c) return GEOSRAD_EINVAL;
    CH_PACK();
    return c->irrad_host(m, np, in, co2, trace, ict, icb, ns, na, nb, aer, out);
}

int geosrad_irrad_dev(geosrad_ctx *c, void *stream, int m, int np, const void *ple, const void *ta, const void *wa, const void *oa,
                      const void *tb, double co2, int trace, const void *n2o, const void *ch4, const void *cfc11, const void *cfc12,
                      const void *cfc22, const void *cwc, const void *fcld, int ict, int icb, const void *reff, int ns, const void *fs,
                      const void *tg, const void *eg, const void *tv, const void *ev, const void *rv, int na, int nb, void *taua,
                      void *ssaa, void *asya, void *flxu, void *flcu, void *flau, void *flxau, void *flxd, void *flcd, void *flad,
                      void *flxad, void *dfdts, void *sfcem, void *taudiag)
{
    if (!c) return GEOSRAD_EINVAL;
    CH_PACK();
    return c->irrad_dev((hipStream_t)stream, m, np, in, co2, trace, ict, icb, ns, na, nb, aer, out);
}

int geosrad_set_tables_chou_sw(geosrad_ctx *c, const void *blob, size_t n) { return c ? c->set_tables_chou_sw(blob, n) : GEOSRAD_EINVAL; }
int geosrad_load_tables_chou_sw(geosrad_ctx *c, const char *path)
{
    if (!c) return GEOSRAD_EINVAL;
    std::vector<char> buf;
    int rc = read_file(c, path, buf);
    return rc ? rc : c->set_tables_chou_sw(buf.data(), buf.size());
}

#define SO_PACK()                                                                                                                    \
    const void *in[SI_NIN] = {cosz, pl, ta, wa, oa, cwc, fcld, reff, taua, ssaa, asya, rsuvbm, rsuvdf, rsirbm, rsirdf};                 \
    void *out[SOO_NOUT] = {flx, flc, fdiruv, fdifuv, fdirpar, fdifpar, fdirir, fdifir, flxu, flcu, flx_sfc_band, drband, dfband}

int geosrad_sorad(geosrad_ctx *c, int m, int np, int nb, const void *cosz, const void *pl, const void *ta, const void *wa, const void *oa,
                  double co2, const void *cwc, const void *fcld, int ict, int icb, const void *reff, const void *hk_uv, const void *hk_ir,
                  const void *taua, const void *ssaa, const void *asya, const void *rsuvbm, const void *rsuvdf, const void *rsirbm,
                  const void *rsirdf, void *flx, void *flc, void *fdiruv, void *fdifuv, void *fdirpar, void *fdifpar, void *fdirir,
                  void *fdifir, void *flxu, void *flcu, void *flx_sfc_band, int do_drfband, void *drband, void *dfband)
{
    if (!c) return GEOSRAD_EINVAL;
    SO_PACK();
    return c->sorad_host(m, np, nb, in, co2, ict, icb, hk_uv, hk_ir, out, do_drfband);
}

int geosrad_sorad_dev(geosrad_ctx *c, void *stream, int m, int np, int nb, const void *cosz, const void *pl, const void *ta, const void *wa,
                      const void *oa, double co2, const void *cwc, const void *fcld, int ict, int icb, const void *reff, const void *hk_uv,
                      const void *hk_ir, const void *taua, const void *ssaa, const void *asya, const void *rsuvbm, const void *rsuvdf,
                      const void *rsirbm, const void *rsirdf, void *flx, void *flc, void *fdiruv, void *fdifuv, void *fdirpar,
                      void *fdifpar, void *fdirir, void *fdifir, void *flxu, void *flcu, void *flx_sfc_band, int do_drfband,
                      void *drband, void *dfband)
{
    if (!c) return GEOSRAD_EINVAL;
    SO_PACK();
    return c->sorad_dev((hipStream_t)stream, m, np, nb, in, co2, ict, icb, hk_uv, hk_ir, out, do_drfband);
}

#define LW_PACK_IN()                                                                                                     \
    const void *in[I_NIN] = {play, plev, tlay, tlev, tsfc, emis, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, \
                             cfc12vmr, cfc22vmr, ccl4vmr, cldf, ciwp, clwp, rei, rel, tauaer, zm, alat}

int geosrad_rrtmg_lw(geosrad_ctx *c, int ncol, int nlay, int psize, int dudTs, const void *play, const void *plev, const void *tlay,
                     const void *tlev, const void *tsfc, const void *emis, const void *h2ovmr, const void *o3vmr, const void *co2vmr,
                     const void *ch4vmr, const void *n2ovmr, const void *o2vmr, const void *cfc11vmr, const void *cfc12vmr,
                     const void *cfc22vmr, const void *ccl4vmr, const void *cldf, const void *ciwp, const void *clwp, const void *rei,
                     const void *rel, int iceflglw, int liqflglw, const void *tauaer, const void *zm, const void *alat, int dyofyr,
                     int cloudLM, int cloudMH, int32_t *clearCounts, void *uflx, void *dflx, void *uflxc, void *dflxc, void *duflx_dTs,
                     void *duflxc_dTs, const int32_t *band_output, void *olrb, void *dolrb_dTs)
{
    if (!c) return GEOSRAD_EINVAL;
    (void)psize;
    LW_PACK_IN();
    void *out[O_NOUT] = {uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs, olrb, dolrb_dTs};
    return c->lw_host(ncol, nlay, dudTs, in, iceflglw, liqflglw, dyofyr, cloudLM, cloudMH, clearCounts, out, band_output, nullptr, nullptr);
}

int geosrad_rrtmg_lw_dev(geosrad_ctx *c, void *stream, int ncol, int nlay, int psize, int dudTs, const void *play, const void *plev,
                         const void *tlay, const void *tlev, const void *tsfc, const void *emis, const void *h2ovmr, const void *o3vmr,
                         const void *co2vmr, const void *ch4vmr, const void *n2ovmr, const void *o2vmr, const void *cfc11vmr,
                         const void *cfc12vmr, const void *cfc22vmr, const void *ccl4vmr, const void *cldf, const void *ciwp,
                         const void *clwp, const void *rei, const void *rel, int iceflglw, int liqflglw, const void *tauaer, const void *zm,
                         const void *alat, int dyofyr, int cloudLM, int cloudMH, int32_t *clearCounts, void *uflx, void *dflx, void *uflxc,
                         void *dflxc, void *duflx_dTs, void *duflxc_dTs, const int32_t *band_output, void *olrb, void *dolrb_dTs)
{
    if (!c || !clearCounts) return GEOSRAD_EINVAL;
    (void)psize;
    LW_PACK_IN();
    void *out[O_NOUT] = {uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs, olrb, dolrb_dTs};
    return c->lw_dev((hipStream_t)stream, ncol, nlay, dudTs, in, iceflglw, liqflglw, dyofyr, cloudLM, cloudMH, clearCounts, out,
                     band_output, nullptr, nullptr, nullptr);
}

int geosrad_rrtmg_lw_rats_dev(geosrad_ctx *c, void *stream, int ncol, int nlay, int psize, int dudTs, const void *play, const void *plev,
                              const void *tlay, const void *tlev, const void *tsfc, const void *emis, const void *h2ovmr, const void *o3vmr,
                              const void *co2vmr, const void *ch4vmr, const void *n2ovmr, const void *o2vmr, const void *cfc11vmr,
                              const void *cfc12vmr, const void *cfc22vmr, const void *ccl4vmr, const void *cldf, const void *ciwp,
                              const void *clwp, const void *rei, const void *rel, int iceflglw, int liqflglw, const void *tauaer,
                              const void *zm, const void *alat, int dyofyr, int cloudLM, int cloudMH, int32_t *clearCounts, void *uflx,
                              void *dflx, void *uflxc, void *dflxc, void *duflx_dTs, void *duflxc_dTs, const int32_t *band_output,
                              void *olrb, void *dolrb_dTs, int nrats, const int32_t *rat_gas, void *uflx_rat, void *dflx_rat,
                              void *duflx_dTs_rat)
{
    if (!c || !clearCounts) return GEOSRAD_EINVAL;
    if (nrats < 0 || nrats > GEOSRAD_RAT_NGAS || (nrats > 0 && !rat_gas)) return c->fail(GEOSRAD_EINVAL, "bad RATS arguments");
    (void)psize;
    LW_PACK_IN();
    void *out[O_NOUT] = {uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs, olrb, dolrb_dTs};
    geosrad_ctx::LwRats RT{};
    RT.n = nrats; RT.uflx = uflx_rat; RT.dflx = dflx_rat; RT.duflx_dTs = duflx_dTs_rat;
    for (int r = 0; r < nrats; r++) RT.gas[r] = rat_gas[r];
    return c->lw_dev((hipStream_t)stream, ncol, nlay, dudTs, in, iceflglw, liqflglw, dyofyr, cloudLM, cloudMH, clearCounts, out,
                     band_output, nullptr, nullptr, nrats > 0 ? &RT : nullptr);
}

int geosrad_lw_driver_rrtmg_dev(geosrad_ctx *c, void *stream, int ncol, int lm, int nb_aer, const void *const *in, const double *consts,
                                int iceflglw, int liqflglw, int doy, int lcldlm, int lcldmh, const int32_t *band_output, void *const *out)
{
    if (!c || !in || !consts || !out) return GEOSRAD_EINVAL;
    return c->lw_driver_dev((hipStream_t)stream, ncol, lm, nb_aer, in, consts, iceflglw, liqflglw, doy, lcldlm, lcldmh, band_output, out, 0,
                            nullptr, nullptr);
}

int geosrad_lw_driver_rrtmg_rats_dev(geosrad_ctx *c, void *stream, int ncol, int lm, int nb_aer, const void *const *in, const double *consts,
                                     int iceflglw, int liqflglw, int doy, int lcldlm, int lcldmh, const int32_t *band_output,
                                     void *const *out, int nrats, const int32_t *rat_gas, void *const *rat_out)
{
    if (!c || !in || !consts || !out) return GEOSRAD_EINVAL;
    return c->lw_driver_dev((hipStream_t)stream, ncol, lm, nb_aer, in, consts, iceflglw, liqflglw, doy, lcldlm, lcldmh, band_output, out,
                            nrats, rat_gas, rat_out);
}

int geosrad_sw_driver_rrtmg_dev(geosrad_ctx *c, void *stream, int ncol, int lm, int nb_aer, const void *const *in, const double *consts,
                                int iceflgsw, int liqflgsw, double sc, double dist, int isolvar, int dyofyr, int include_aerosols,
                                int lcldlm, int lcldmh, int normflx, const void *bndsolvar, const void *indsolvar, void *const *out)
{
    if (!c || !in || !consts || !out) return GEOSRAD_EINVAL;
    return c->sw_driver_dev((hipStream_t)stream, ncol, lm, nb_aer, in, consts, iceflgsw, liqflgsw, sc, dist, isolvar, dyofyr,
                            include_aerosols, lcldlm, lcldmh, normflx, bndsolvar, indsolvar, out);
}

int geosrad_sw_driver_chou_dev(geosrad_ctx *c, void *stream, int ncol, int lm, const void *const *in, const double *consts, int lcldmh,
                               int lcldlm, const void *hk_uv, const void *hk_ir, int do_drfband, void *const *out)
{
    if (!c || !in || !out) return GEOSRAD_EINVAL;
    return c->sw_driver_chou_dev((hipStream_t)stream, ncol, lm, in, consts, lcldmh, lcldlm, hk_uv, hk_ir, do_drfband, out);
}

int geosrad_lw_chou_post_dev(geosrad_ctx *c, void *stream, int ncol, int lm, const void *const *in, void *const *out)
{
    if (!c || !in || !out) return GEOSRAD_EINVAL;
    return c->lw_chou_post_dev((hipStream_t)stream, ncol, lm, in, out);
}

int geosrad_lw_update_flx_dev(geosrad_ctx *c, void *stream, int ncol, int lm, int rrtmg, int lev_mid_high, int lev_low_mid, double undef,
                              const void *const *in, void *const *out)
{
    if (!c || !in || !out) return GEOSRAD_EINVAL;
    return c->lw_update_flx_dev((hipStream_t)stream, ncol, lm, rrtmg, lev_mid_high, lev_low_mid, undef, in, out);
}

int geosrad_lw_update_rats_dev(geosrad_ctx *c, void *stream, int ncol, int lm, int nrats, const void *const *in, void *const *out)
{
    if (!c || !in || !out) return GEOSRAD_EINVAL;
    return c->lw_update_rats_dev((hipStream_t)stream, ncol, lm, nrats, in, out);
}

int geosrad_lw_update_bands_dev(geosrad_ctx *c, void *stream, int ncol, const int32_t *band_output, const double *wavenum1,
                                const double *wavenum2, double undef, const void *tsinst, const void *ts_int, const void *olrb_int,
                                const void *dolrb_int, void *olrb_exp, void *tbrb_exp)
{
    if (!c) return GEOSRAD_EINVAL;
    return c->lw_update_bands_dev((hipStream_t)stream, ncol, band_output, wavenum1, wavenum2, undef, tsinst, ts_int, olrb_int, dolrb_int,
                                  olrb_exp, tbrb_exp);
}

int geosrad_sw_update_export_dev(geosrad_ctx *c, void *stream, int ncol, int lm, int nbands, const void *const *in, void *const *out)
{
    if (!c || !in || !out) return GEOSRAD_EINVAL;
    return c->sw_update_export_dev((hipStream_t)stream, ncol, lm, nbands, in, out);
}

int geosrad_sw_update_surface_dev(geosrad_ctx *c, void *stream, int ncol, int lm, double undef, const void *const *in, void *const *out)
{
    if (!c || !in || !out) return GEOSRAD_EINVAL;
    return c->sw_update_surface_dev((hipStream_t)stream, ncol, lm, undef, in, out);
}

int geosrad_rad_tendencies_dev(geosrad_ctx *c, void *stream, int ncol, int lm, double grav, double cp, const void *const *in,
                               void *const *out)
{
    if (!c || !in || !out) return GEOSRAD_EINVAL;
    return c->rad_tendencies_dev((hipStream_t)stream, ncol, lm, grav, cp, in, out);
}

int geosrad_lit_index_dev(geosrad_ctx *c, void *stream, int ncol, const void *zth, int32_t *lit_index, int32_t *lit_pos, int32_t *nlit_dev,
                          int *nlit_host)
{
    return c ? c->lit_index_dev((hipStream_t)stream, ncol, zth, lit_index, lit_pos, nlit_dev, nlit_host) : GEOSRAD_EINVAL;
}
int geosrad_lit_pack_dev(geosrad_ctx *c, void *stream, int pdim, int udim, int nlev, const int32_t *lit_index, const int32_t *nlit_dev,
                         const void *unpacked, void *packed)
{
    return c ? c->lit_pack_dev((hipStream_t)stream, pdim, udim, nlev, lit_index, nlit_dev, unpacked, packed) : GEOSRAD_EINVAL;
}
int geosrad_lit_unpack_dev(geosrad_ctx *c, void *stream, int pdim, int udim, int nlev, const int32_t *lit_pos, const void *packed,
                           void *unpacked, int use_default, double dflt)
{
    return c ? c->lit_unpack_dev((hipStream_t)stream, pdim, udim, nlev, lit_pos, packed, unpacked, use_default, dflt) : GEOSRAD_EINVAL;
}

int geosrad_profile(geosrad_ctx *c, int enable)
{
    if (!c) return GEOSRAD_EINVAL;
    c->prof_collect();
    c->profiling = enable != 0;
    for (int k = 0; k < 16; k++) { c->prof_ms[k] = 0; c->prof_n[k] = 0; }
    return GEOSRAD_OK;
}
int geosrad_profile_read(geosrad_ctx *c, int kernel_id, double *total_ms, long *launches)
{
    if (!c || kernel_id < 0 || kernel_id >= 16) return GEOSRAD_EINVAL;
    (void)hipSetDevice(c->device);
    c->prof_collect();
    if (total_ms) *total_ms = c->prof_ms[kernel_id];
    if (launches) *launches = c->prof_n[kernel_id];
    return GEOSRAD_OK;
}
const char *geosrad_kernel_name(int kernel_id)
{
    static const char *nm[14] = {"k_validate_pwv", "k_setcoef", "k_overlap", "k_mcica", "k_lw_bands", "k_lw_reduce",
                                 "k_sw_validate", "k_sw_setcoef", "k_sw_bands", "k_sw_reduce", "k_chou_prep", "k_chou_bands",
                                 "k_sorad_prep", "k_sorad_pass"};
    return kernel_id >= 0 && kernel_id < 14 ? nm[kernel_id] : "";
}

// the name of the kernel that runs in a profile slot under THIS context's kernel paths (GEOSRAD_SW_PATH / _LW_PATH / _SORAD_PATH), as it
// appears in rocprofv3's kernel trace
const char *geosrad_kernel_label(geosrad_ctx *c, int kernel_id)
{
    if (!c) return geosrad_kernel_name(kernel_id);
    switch (kernel_id) {
    case 4: return c->lw_cols_path ? "k_lw_cols" : (c->lw_split_path ? "k_lw_cells+k_lw_sweep" : "k_lw_bands");
    case 8: return c->sw_path == 2 ? "k_sw_reform" : "k_sw_bands";
    case 9: return c->sw_path == 2 ? "k_swr_reduce" : "k_sw_reduce";
    case 13: return c->sorad_col_path ? "k_sorad_col" : "k_sorad_pass";
    default: return geosrad_kernel_name(kernel_id);
    }
}

int geosrad_check(geosrad_ctx *c, void *stream) { return c ? c->check((hipStream_t)stream) : GEOSRAD_EINVAL; }

// test hook: gr_div64 / gr_rcp64 / gr_sqrt64 (lw_kernels.hpp), the division / reciprocal / square root of the fp64 RRTMG_SW instantiation
static __global__ void k_dbg_fast64(int n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ q,
                                    double *__restrict__ r, double *__restrict__ s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { q[i] = geosrad::gr_div64(a[i], b[i]); r[i] = geosrad::gr_rcp64(b[i]); s[i] = geosrad::gr_sqrt64(b[i]); }
}
int geosrad_dbg_fast64(int n, const double *a, const double *b, double *quot, double *rcp, double *root)
{
    if (n <= 0 || !a || !b || !quot || !rcp || !root) return GEOSRAD_EINVAL;
    double *d = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    if (hipMalloc(&d, 5 * nb) != hipSuccess) return GEOSRAD_EHIP;
    int rc = GEOSRAD_OK;
    if (hipMemcpy(d, a, nb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d + n, b, nb, hipMemcpyHostToDevice) != hipSuccess) rc = GEOSRAD_EHIP;
    if (rc == GEOSRAD_OK) {
        k_dbg_fast64<<<(n + 255) / 256, 256>>>(n, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, d + 4 * (size_t)n);
        if (hipMemcpy(quot, d + 2 * (size_t)n, nb, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(rcp, d + 3 * (size_t)n, nb, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(root, d + 4 * (size_t)n, nb, hipMemcpyDeviceToHost) != hipSuccess) rc = GEOSRAD_EHIP;
    }
    (void)hipFree(d);
    return rc;
}

int geosrad_rrtmg_lw_taumol(geosrad_ctx *c, int ncol, int nlay, const void *play, const void *plev, const void *tlay, const void *tlev,
                            const void *tsfc, const void *emis, const void *h2ovmr, const void *o3vmr, const void *co2vmr,
                            const void *ch4vmr, const void *n2ovmr, const void *o2vmr, const void *cfc11vmr, const void *cfc12vmr,
                            const void *cfc22vmr, const void *ccl4vmr, const void *tauaer, void *taug, void *pfracs)
{
    if (!c || !taug || !pfracs) return GEOSRAD_EINVAL;
    // clear-sky run with a zero cloud field; fluxes are discarded
    const size_t esz = (size_t)c->real_kind;
    std::vector<char> zero((size_t)ncol * (nlay + 1) * esz, 0), ones((size_t)ncol * nlay * esz, 0), scratch((size_t)ncol * (nlay + 1) * esz * 4);
    for (size_t i = 0; i < (size_t)ncol * nlay; i++) { if (esz == 4) ((float *)ones.data())[i] = 10.f; else ((double *)ones.data())[i] = 10.; }
    std::vector<int32_t> cc((size_t)ncol * 4);
    const void *cldf = zero.data(), *ciwp = zero.data(), *clwp = zero.data(), *rei = ones.data(), *rel = ones.data(), *zm = zero.data(),
               *alat = zero.data();
    LW_PACK_IN();
    char *s = scratch.data();
    const size_t cv = (size_t)ncol * (nlay + 1) * esz;
    void *out[O_NOUT] = {s, s + cv, s + 2 * cv, s + 3 * cv, nullptr, nullptr, nullptr, nullptr};
    return c->lw_host(ncol, nlay, 0, in, 3, 1, 1, 1, 2, cc.data(), out, nullptr, taug, pfracs);
}

int geosrad_mcica(geosrad_ctx *c, int ncol, int nsubcol, int nlay, const void *zmid, const void *alat, int doy, const void *play,
                  const void *cldfrac, const void *ciwp, const void *clwp, double cwp_tiny, const int32_t seed_order[4],
                  int32_t *cldy_stoch, void *ciwp_stoch, void *clwp_stoch)
{
    if (!c || !zmid || !alat || !play || !cldfrac || !ciwp || !clwp || !cldy_stoch || !ciwp_stoch || !clwp_stoch) return GEOSRAD_EINVAL;
    return c->mcica_host(ncol, nsubcol, nlay, zmid, alat, doy, play, cldfrac, ciwp, clwp, cwp_tiny, seed_order, cldy_stoch, ciwp_stoch,
                         clwp_stoch);
}

int geosrad_mcica_dev(geosrad_ctx *c, void *stream, int ncol, int nsubcol, int nlay, const void *zmid, const void *alat, int doy,
                      const void *play, const void *cldfrac, const void *ciwp, const void *clwp, double cwp_tiny,
                      const int32_t seed_order[4], int32_t *cldy_stoch, void *ciwp_stoch, void *clwp_stoch)
{
    if (!c || !zmid || !alat || !play || !cldfrac || !ciwp || !clwp || !cldy_stoch || !ciwp_stoch || !clwp_stoch) return GEOSRAD_EINVAL;
    return c->mcica_dev((hipStream_t)stream, ncol, nsubcol, nlay, zmid, alat, doy, play, cldfrac, ciwp, clwp, cwp_tiny, seed_order,
                        cldy_stoch, ciwp_stoch, clwp_stoch);
}

int geosrad_clearcounts(geosrad_ctx *c, int ncol, int nsubcol, int nlay, int cloudLM, int cloudMH, const int32_t *cldy, int32_t *cnt)
{
    if (!c || !cldy || !cnt || ncol <= 0) return GEOSRAD_EINVAL;
    if (cloudLM == cloudMH) return c->fail(GEOSRAD_EINPUT, "invalid pressure super-layers!");
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(GEOSRAD_EHIP, "hipSetDevice");
    int32_t *d_in = nullptr, *d_out = nullptr;
    const size_t nin = (size_t)ncol * nsubcol * nlay * 4;
    if (hipMalloc((void **)&d_in, nin) != hipSuccess || hipMalloc((void **)&d_out, (size_t)ncol * 16) != hipSuccess) {
        if (d_in) (void)hipFree(d_in);
        return c->fail(GEOSRAD_ENOMEM, "hipMalloc failed in geosrad_clearcounts");
    }
    int rc = GEOSRAD_OK;
    if (hipMemcpy(d_in, cldy, nin, hipMemcpyHostToDevice) != hipSuccess) rc = c->fail(GEOSRAD_EHIP, "hipMemcpy H2D");
    if (!rc) {
        hipLaunchKernelGGL(k_clearcounts, dim3((unsigned)((ncol + 63) / 64)), dim3(64), 0, c->stream, ncol, nsubcol, nlay, cloudLM, cloudMH,
                           (const int32_t *)d_in, d_out);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(cnt, d_out, (size_t)ncol * 16, hipMemcpyDeviceToHost) != hipSuccess)
            rc = c->fail(GEOSRAD_EHIP, "k_clearcounts failed");
    }
    (void)hipFree(d_in); (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
#endif   // GEOSRAD_PART == 0
