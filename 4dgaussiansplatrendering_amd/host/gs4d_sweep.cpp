// gs4d_sweep.cpp — the multi-GPU leg of the path driven from C++: BASELINE.json configs[3], the 256-frame time sweep of 10^6 4D splats
// (t_k = 50 k / 255), frame k on rank k mod N, one process per GPU, RGBA8 frames gathered on rank 0 over RCCL.
//
// The reference renders its sweep in one process with one GL context (Application.cpp:145-190 calls Scene::Update / Render per frame,
// Scenes.h:289-340); frames are independent, so the sweep shards by frame with no data-path collective — the only exchange is the
// presentation: every rank hands its frames to the rank that owns the display.  bench.py's N > 1 leg does the same from Python with
// torch.distributed; this is the same schedule with nothing but the C ABI (include/gs4d.h), HIP and RCCL.
//
//   gs4d_sweep --gpus N            forks N ranks itself (before anything touches the GPU), rank r on device r
//   RANK=r WORLD_SIZE=N LOCAL_RANK=l gs4d_sweep      one rank, started by a launcher of your own (MASTER_PORT names the rendezvous file)
//
//   gs4d_sweep --gpus N --shard-tiles   BASELINE.json configs[4]'s shape instead: ONE frame per step, its rows of 8x8 tiles dealt round-robin
//                                  to the ranks (gs4d_set_tile_shard; every rank sorts all splats, composites its rows), bands gathered on rank 0
//
// Rendezvous: rank 0 writes {job nonce, ncclUniqueId} to a file under /tmp (renamed into place), the others wait for a file carrying
// THEIR job's nonce — a file left behind by an earlier, failed job is ignored (the nonce is the launcher's run id where there is one;
// without one, a file older than the waiting process is).
// Presentation is software-pipelined as a swap chain is (frame j is queued first, then frame j-1 — the previous image — is packed to
// RGBA8 into the batch) and every G presented frames the batch goes to rank 0: ncclSend / ncclRecv in one group, on a stream of this
// program's that the context knows as the caller's stream (gs4d_set_stream).  There are TWO batch buffers, used alternately: a pack
// into one waits only for the gather that last read THAT buffer (an event of this program's, handed to
// gs4d_read_frame_rgba8_device_after), never for the gather of the other buffer that is still in flight; a gather waits for the packs it
// sends (inside the library).  Every rank presents ceil(256 / N) times, so all ranks make the same collective calls whatever the world size.
// The line reports, per rank, how long the comm stream was busy per sweep (sum of its gathers: for a sender that includes waiting for
// rank 0 to post the matching receive) and how long it ran on after the rank's last frame had been rendered (the exposed tail).
#include "../../include/gs4d.h"

#include <hip/hip_runtime_api.h>
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

namespace {

struct Args {
    int gpus = 1, frames = 256, gather_every = 8, sweeps = 3, warmup = 1, width = 1920, height = 1080;
    int fake_comm_us = 0;        // experiment (needs a TUNING build of libgs4d.so): every gather also occupies the comm stream for this long per frame it sends
    int comm_priority = -1;      // the comm stream's priority class: 0 default, 1 highest, 2 lowest; -1 = lowest when there is more than one rank, default otherwise
    int batch_buffers = 2;       // batch buffers used in turn (2..8)
    int rank0_pct = -1;          // rank 0 renders this percentage of an equal share of the sweep (it also receives every other rank's frames); -1: the model of deal_default_pct()
    size_t splats = 1000000;
    float t_max = 50.0f;
    std::string dump;            // directory: rank 0 writes records.bin and frame_####.rgba8 of the verification sweep (tests)
    std::string png;             // prefix: rank 0 writes <prefix>####.png for every --png-every-th frame of the verification sweep
    int png_every = 32;
    bool verify = true;          // one untimed sweep whose frames are copied to the host on rank 0 and check-summed in frame order
    bool shard_tiles = false;    // one frame per step, tile rows sharded over the ranks (configs[4]) instead of frames (configs[3])
};

#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLOK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, ncclGetErrorString(r_)); return 1; } } while (0)
#define GSOK(call) do { int r_ = (call); if (r_ != GS4D_OK) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, gs4d_last_error(ctx)); return 1; } } while (0)
int g_rank = 0;
long g_started = 0;              // wall-clock second this process entered main()

// ---- synthetic scene: the counter-based generator of tests/scenes.py (splitmix64 of seed ^ (index * 64 + stream) -> 24-bit uniforms) ----
uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
double uniform(uint64_t i, int stream) { return (double)(splitmix64(0x4D495335ull ^ (i * 64ull + (uint64_t)stream)) >> 40) / 16777216.0; }
double normal(uint64_t i, int stream) {
    const double u1 = uniform(i, stream), u2 = uniform(i, stream + 1);
    return std::sqrt(-2.0 * std::log(std::fmax(u1, std::ldexp(1.0, -25)))) * std::cos(2.0 * M_PI * u2);
}

// C4 of SURVEY.md 8(d): mu ~ U[-200,200]^3, mu_t ~ U[0,50], q = normalised N(0,1)^4, scale ~ U[0.5,2]^3, lifetime ~ U[0.5,2], fade 0.5,
// velocity ~ U[-5,5]^3, rgb ~ U[0,1]^3, alpha ~ U[0.2,1]; the 96-byte records are built by the library's own Splat4D constructor
// (gs4d_host_build_records_4d, Splat.h:132-159).
void make_records(size_t n, std::vector<float>& rec) {
    std::vector<float> pos4(4 * n), q(4 * n), sc(3 * n), life(n), fade(n, 0.5f), vel(3 * n), col(4 * n);
    for (size_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) pos4[4 * i + a] = (float)(uniform(i, a) * 400.0 - 200.0);
        pos4[4 * i + 3] = (float)(uniform(i, 18) * 50.0);
        double qq[4] = { normal(i, 3), normal(i, 5), normal(i, 7), normal(i, 9) };
        const double len = std::sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
        for (int a = 0; a < 4; ++a) q[4 * i + a] = (float)(qq[a] / len);
        for (int a = 0; a < 3; ++a) sc[3 * i + a] = (float)(uniform(i, 11 + a) * 1.5 + 0.5);
        life[i] = (float)(uniform(i, 19) * 1.5 + 0.5);
        for (int a = 0; a < 3; ++a) vel[3 * i + a] = (float)(uniform(i, 20 + a) * 10.0 - 5.0);
        for (int a = 0; a < 3; ++a) col[4 * i + a] = (float)uniform(i, 14 + a);
        col[4 * i + 3] = (float)(uniform(i, 17) * 0.8 + 0.2);
    }
    rec.resize(24 * n);
    gs4d_host_build_records_4d(n, pos4.data(), q.data(), sc.data(), life.data(), fade.data(), vel.data(), col.data(), rec.data());
}

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 255u] ^ (crc >> 8);
    return ~crc;
}

bool write_file(const std::string& path, const void* p, size_t bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(p, 1, bytes, f) == bytes;
    fclose(f);
    return ok;
}

struct Rendezvous { char nonce[64]; ncclUniqueId id; };

int sweep_tiles(const Args& a, int rank, int world, gs4d_ctx* ctx, ncclComm_t comm, hipStream_t stream, gs4d_buf data, const std::vector<gs4d_buf>& keys, const std::vector<gs4d_buf>& idx, int lanes, const float cam_pos[3]);

int run_rank(const Args& a, int rank, int world, int local_rank, const std::string& idfile, const std::string& nonce) {
    g_rank = rank;
    gs4d_ctx* ctx = nullptr;
    HIPOK(hipSetDevice(local_rank));
    // The context first, the communicator and this program's own stream after it: the frame lanes' streams are then the first streams the process
    // creates (streams alive at context creation change how HIP maps the lanes onto hardware queues: tools/order_effect.py).
    if (gs4d_create(local_rank, a.width, a.height, &ctx) != GS4D_OK) { fprintf(stderr, "[rank %d] gs4d_create: %s\n", rank, gs4d_last_error(nullptr)); return 1; }
    // ---- communicator ----
    Rendezvous rv;
    memset(&rv, 0, sizeof rv);
    if (rank == 0) {
        unlink(idfile.c_str());                                              // whatever an earlier job left there
        NCCLOK(ncclGetUniqueId(&rv.id));
        snprintf(rv.nonce, sizeof rv.nonce, "%s", nonce.c_str());
        const std::string tmp = idfile + ".tmp";
        if (!write_file(tmp, &rv, sizeof rv) || rename(tmp.c_str(), idfile.c_str()) != 0) { fprintf(stderr, "cannot write %s\n", idfile.c_str()); return 1; }
    } else {
        bool got = false;
        for (int tries = 0; tries < 6000 && !got; ++tries) {                  // up to 60 s
            FILE* f = fopen(idfile.c_str(), "rb");
            if (f) {
                // another job's file — a different nonce, or (no job id to compare: nonce empty) a file older than this process: keep waiting
                struct stat sb;
                got = fread(&rv, 1, sizeof rv, f) == sizeof rv && (nonce.empty() ? (fstat(fileno(f), &sb) == 0 && (long)sb.st_mtime >= g_started - 10) : strncmp(rv.nonce, nonce.c_str(), sizeof rv.nonce) == 0);
                fclose(f);
            }
            if (!got) usleep(10000);
        }
        if (!got) { fprintf(stderr, "[rank %d] no rendezvous file %s for job %s\n", rank, idfile.c_str(), nonce.c_str()); return 1; }
    }
    const ncclUniqueId id = rv.id;
    ncclComm_t comm;
    NCCLOK(ncclCommInitRank(&comm, world, id, rank));
    // HIP keeps a pool of hardware queues per priority class, and a queue executes in order: in the default class the communication stream shares a
    // queue with one of the frame lanes, and a send that holds it for a millisecond stalls that lane — and, through the host's validation waits, every
    // lane.  Emulated on one GPU (--fake-comm-us 150: the stream is held 150 us per frame sent, what 8.3 MB take on one xGMI link): 68.6 ms per sweep
    // in the default class, 53.3 in the highest, 46.5 in the lowest (rendering alone: 38.0; without any traffic the lowest class costs 40.5).
    const int comm_priority = a.comm_priority >= 0 ? a.comm_priority : (world > 1 ? 2 : 0);
    hipStream_t stream;
    if (comm_priority) { int least = 0, greatest = 0; HIPOK(hipDeviceGetStreamPriorityRange(&least, &greatest)); HIPOK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, comm_priority == 2 ? least : greatest)); }
    else HIPOK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    typedef int (*spin_fn)(void*, unsigned);
    spin_fn fake_spin = a.fake_comm_us ? (spin_fn)dlsym(RTLD_DEFAULT, "gs4d_tuning_spin") : nullptr;
    if (a.fake_comm_us && !fake_spin) { fprintf(stderr, "--fake-comm-us needs a tuning build of libgs4d.so (make lib TUNING=1)\n"); return 1; }

    // ---- scene, resident on the device ----
    const size_t n = a.splats;
    std::vector<float> rec;
    make_records(n, rec);
    GSOK(gs4d_set_stream(ctx, stream));
    uint64_t st[8];
    GSOK(gs4d_get_stats(ctx, st));
    const int lanes = (int)(st[6] & 0xFFFFu);
    gs4d_buf data = 0;
    GSOK(gs4d_buffer_create(ctx, rec.data(), rec.size() * 4, &data));
    std::vector<gs4d_buf> keys(lanes), idx(lanes);
    for (int l = 0; l < lanes; ++l) { GSOK(gs4d_buffer_create(ctx, nullptr, n * 4, &keys[l])); GSOK(gs4d_buffer_create(ctx, nullptr, n * 4, &idx[l])); }
    const float cam_pos[3] = { 551.58f, 350.43f, -184.33f }, cam_dir[3] = { -0.774978f, -0.570354f, 0.272222f }, up[3] = { 0.0f, 1.0f, 0.0f };   // the README screenshot camera (tests/scenes.py CAM_CUBE)
    float view[16], proj[16];
    gs4d_host_look_at(cam_pos, cam_dir, up, view);
    gs4d_host_perspective(60.0f, a.width, a.height, 0.1f, 5000.0f, proj);                 // Camera.h:71-73, Application.cpp:126
    const float clear[4] = { 0.1843137254901961f, 0.20784313725490197f, 0.25882352941176473f, 1.0f };      // Application.cpp:125
    GSOK(gs4d_set_clear_color(ctx, clear));
    GSOK(gs4d_set_mode(ctx, GS4D_MODE_4D_SORTED));
    GSOK(gs4d_bind_storage(ctx, 2, data));
    GSOK(gs4d_set_uniform_1f(ctx, GS4D_U_MIN_OPACITY, 0.0f));
    GSOK(gs4d_set_uniform_mat4(ctx, GS4D_U_VIEW, view));
    GSOK(gs4d_set_uniform_mat4(ctx, GS4D_U_PROJ, proj));
    if (a.shard_tiles) {
        const int rc = sweep_tiles(a, rank, world, ctx, comm, stream, data, keys, idx, lanes, cam_pos);
        gs4d_destroy(ctx);
        (void)hipStreamDestroy(stream);
        ncclCommDestroy(comm);
        if (rank == 0) unlink(idfile.c_str());
        return rc;
    }

    // ---- frames of this rank, batches ----
    // The deal (sharding.py deal()): pct = 100 -> frame k on rank k mod world.  Otherwise rank 0 — which also receives every other rank's frames:
    // (world - 1) x ~54 GB/s written into its HBM while it renders, out of the ~3 TB/s its own frames' traffic achieves — renders
    // round(frames / world * pct / 100) frames, spread evenly over the sweep, and the rest go round-robin to ranks 1..world-1.
    const int pct = a.rank0_pct >= 0 ? a.rank0_pct : std::max(10, (int)std::lround(100.0 * (1.0 - (world - 1) * 54.0 / 3000.0)));
    std::vector<std::vector<int>> frames_of(world);
    if (world <= 1 || pct == 100) { for (int k = 0; k < a.frames; ++k) frames_of[world <= 1 ? 0 : k % world].push_back(k); }
    else {
        const long n0 = std::min<long>(a.frames, std::max<long>(0, std::lround((double)a.frames / world * pct / 100.0)));
        int nxt = 0;
        for (long k = 0; k < a.frames; ++k) {
            if ((k + 1) * n0 / a.frames > k * n0 / a.frames) frames_of[0].push_back((int)k);
            else frames_of[1 + nxt++ % (world - 1)].push_back((int)k);
        }
    }
    const std::vector<int>& mine = frames_of[rank];
    int most = 0;                                                            // presentations per rank and sweep: the busiest rank's frames
    for (const auto& f : frames_of) most = std::max(most, (int)f.size());
    const int G = a.gather_every < 1 ? 1 : a.gather_every;
    const size_t fbytes = (size_t)a.width * a.height * 4;
    // NB batch buffers, used in turn (gather b reads buffer b % NB while the packs of the following batches fill the others)
    const int NB = a.batch_buffers;
    uint8_t* batch[8] = { nullptr }; uint8_t* gathered[8] = { nullptr }; double* dmax = nullptr;
    hipEvent_t ev_free[8]; bool ev_valid[8] = { false };                      // recorded behind the gather that last read buffer x
    for (int x = 0; x < NB; ++x) {
        if (rank == 0) { HIPOK(hipMalloc(&gathered[x], (size_t)world * G * fbytes)); batch[x] = gathered[x]; }      // rank 0 packs straight into its slice of the gathered batch
        else HIPOK(hipMalloc(&batch[x], G * fbytes));
        HIPOK(hipMemset(batch[x], 0, G * fbytes));
        HIPOK(hipEventCreateWithFlags(&ev_free[x], hipEventDisableTiming));
    }
    HIPOK(hipMalloc(&dmax, sizeof(double) * (size_t)(2 * world + 2)));
    HIPOK(hipMemset(dmax, 0, sizeof(double) * (size_t)(2 * world + 2)));
    // comm-stream accounting: a pair of timing events around every gather of a sweep
    // a gather after every full batch and, inside the last batch of a sweep, after every quarter of one (sharding.py gather_schedule: what is on
    // the wire when a rank's last frame has been rendered is then a quarter of a batch)
    const int piece = G / 4 < 1 ? 1 : G / 4, last_batch = most > 0 ? (most - 1) / G : 0;
    const int max_gathers = (most + G - 1) / G + G / piece + 1;
    std::vector<hipEvent_t> g0(max_gathers), g1(max_gathers);
    for (int i = 0; i < max_gathers; ++i) { HIPOK(hipEventCreate(&g0[i])); HIPOK(hipEventCreate(&g1[i])); }
    int gathers_this_sweep = 0;
    std::vector<uint8_t> host_frames;                                       // verification sweep only, rank 0
    std::vector<uint32_t> frame_crc(a.frames, 0u);
    const bool pipelined = lanes >= 2;
    uint64_t frame_no = 0;

    auto frame = [&](int k) -> int {
        const float t = a.frames > 1 ? a.t_max * (float)k / (float)(a.frames - 1) : 0.0f;
        const int b = (int)(frame_no++ % (uint64_t)lanes);
        GSOK(gs4d_clear(ctx));
        GSOK(gs4d_set_uniform_1f(ctx, GS4D_U_TIME, t));
        GSOK(gs4d_keygen(ctx, data, t, cam_pos, keys[b], idx[b], n, GS4D_KEY_REF_INV_EUCLID));
        GSOK(gs4d_sort_pairs(ctx, keys[b], idx[b], n));
        GSOK(gs4d_bind_storage(ctx, 1, idx[b]));
        GSOK(gs4d_draw_instanced(ctx, n));
        return 0;
    };
    auto gather = [&](int batch_no, int lo, int hi, bool verify) -> int {      // slots [lo, hi) of batch `batch_no`
        const int x = batch_no % NB;
        const size_t off = (size_t)lo * fbytes, len = (size_t)(hi - lo) * fbytes;
        const int gi = gathers_this_sweep < max_gathers ? gathers_this_sweep : max_gathers - 1;
        HIPOK(hipEventRecord(g0[gi], stream));
        NCCLOK(ncclGroupStart());
        if (rank == 0) { for (int r = 1; r < world; ++r) NCCLOK(ncclRecv(gathered[x] + (size_t)r * G * fbytes + off, len, ncclUint8, r, comm, stream)); }
        else NCCLOK(ncclSend(batch[x] + off, len, ncclUint8, 0, comm, stream));
        NCCLOK(ncclGroupEnd());
        if (fake_spin && fake_spin((void*)stream, (unsigned)(a.fake_comm_us * (hi - lo))) != 0) return 1;      // stand-in for the time a real send holds the stream
        HIPOK(hipEventRecord(g1[gi], stream));
        ++gathers_this_sweep;
        HIPOK(hipEventRecord(ev_free[x], stream));                            // buffer x may be packed into again behind this
        ev_valid[x] = true;
        if (verify && rank == 0) {
            host_frames.resize((size_t)world * G * fbytes);
            HIPOK(hipMemcpyAsync(host_frames.data(), gathered[x], host_frames.size(), hipMemcpyDeviceToHost, stream));
            HIPOK(hipStreamSynchronize(stream));
            for (int r = 0; r < world; ++r)
                for (int p = lo; p < hi; ++p) {
                    const int pn = batch_no * G + p;                          // slot p of rank r's batch holds its pn-th frame
                    if (pn >= (int)frames_of[r].size()) continue;
                    const int k = frames_of[r][pn];
                    const uint8_t* f = host_frames.data() + ((size_t)r * G + p) * fbytes;
                    frame_crc[k] = crc32_update(0u, f, fbytes);
                    char name[64];
                    if (!a.dump.empty()) { snprintf(name, sizeof name, "/frame_%04d.rgba8", k); if (!write_file(a.dump + name, f, fbytes)) return 1; }
                    if (!a.png.empty() && k % a.png_every == 0) { snprintf(name, sizeof name, "%04d.png", k); gs4d_host_write_png((a.png + name).c_str(), f, a.width, a.height); }
                }
        }
        return 0;
    };
    auto sweep = [&](bool verify) -> int {
        int presented = 0, slot_lo = 0;
        gathers_this_sweep = 0;
        auto present = [&](int j, int frames_back) -> int {
            const int x = (presented / G) % NB;
            if (j < (int)mine.size()) GSOK(gs4d_read_frame_rgba8_device_after(ctx, frames_back, batch[x] + (size_t)(presented % G) * fbytes, fbytes, ev_valid[x] ? (void*)ev_free[x] : nullptr));
            ++presented;
            const int b = (presented - 1) / G, hi = (presented - 1) % G + 1;
            if (presented % G == 0 || presented == most || (b == last_batch && hi % piece == 0)) {
                if (gather(b, slot_lo, hi, verify)) return 1;
                slot_lo = presented % G == 0 ? 0 : hi;
            }
            return 0;
        };
        for (int j = 0; j < most; ++j) {
            const bool rendered = j < (int)mine.size();
            if (rendered && frame(mine[j])) return 1;
            if (!pipelined) { if (present(j, 0)) return 1; }
            else if (j >= 1) { if (present(j - 1, rendered ? 1 : 0)) return 1; }   // no new frame was started: frame j-1 is still the current image
        }
        if (pipelined && present(most - 1, 0)) return 1;                    // the last frame of the sweep is presented inside the sweep
        return 0;
    };
    double comm_tail_s = 0.0;
    auto fence = [&]() -> int {                                             // everything queued is done on every rank
        GSOK(gs4d_finish(ctx));
        const auto tf = std::chrono::steady_clock::now();
        HIPOK(hipStreamSynchronize(stream));
        comm_tail_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - tf).count();      // how long the comm stream ran on after the last frame was rendered
        NCCLOK(ncclAllReduce(dmax, dmax, 1, ncclDouble, ncclMax, comm, stream));
        HIPOK(hipStreamSynchronize(stream));
        return 0;
    };

    if (rank == 0 && !a.dump.empty()) { mkdir(a.dump.c_str(), 0755); if (!write_file(a.dump + "/records.bin", rec.data(), rec.size() * 4)) return 1; }
    if (a.verify) { if (sweep(true) || fence()) return 1; }
    for (int w = 0; w < a.warmup; ++w) { if (sweep(false)) return 1; }
    if (fence()) return 1;
    std::vector<double> secs;
    double comm_busy_ms = 0.0, comm_tail_ms = 0.0;                          // of this rank, averaged over the timed sweeps
    for (int s = 0; s < a.sweeps; ++s) {
        const auto t0 = std::chrono::steady_clock::now();
        if (sweep(false) || fence()) return 1;
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < gathers_this_sweep && i < max_gathers; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, g0[i], g1[i]) == hipSuccess) comm_busy_ms += ms / a.sweeps; }
        comm_tail_ms += comm_tail_s * 1e3 / a.sweeps;
        HIPOK(hipMemcpy(dmax, &el, sizeof el, hipMemcpyHostToDevice));     // maximum over ranks
        NCCLOK(ncclAllReduce(dmax, dmax, 1, ncclDouble, ncclMax, comm, stream));
        HIPOK(hipStreamSynchronize(stream));
        HIPOK(hipMemcpy(&el, dmax, sizeof el, hipMemcpyDeviceToHost));
        secs.push_back(el);
    }
    GSOK(gs4d_get_stats(ctx, st));
    // every rank's comm accounting on rank 0 (two doubles per rank)
    std::vector<double> comm_all((size_t)2 * world, 0.0);
    {
        const double mine2[2] = { comm_busy_ms, comm_tail_ms };
        HIPOK(hipMemcpy(dmax + 2, mine2, sizeof mine2, hipMemcpyHostToDevice));
        double* all = nullptr;
        HIPOK(hipMalloc(&all, sizeof(double) * 2 * (size_t)world));
        NCCLOK(ncclAllGather(dmax + 2, all, 2, ncclDouble, comm, stream));
        HIPOK(hipStreamSynchronize(stream));
        HIPOK(hipMemcpy(comm_all.data(), all, sizeof(double) * 2 * (size_t)world, hipMemcpyDeviceToHost));
        (void)hipFree(all);
    }
    if (rank == 0) {
        uint32_t crc = 0u;
        for (int k = 0; k < a.frames; ++k) crc = crc32_update(crc, (const uint8_t*)&frame_crc[k], 4);
        std::vector<double> sorted = secs;
        for (size_t i = 0; i < sorted.size(); ++i) for (size_t j = i + 1; j < sorted.size(); ++j) if (sorted[j] < sorted[i]) std::swap(sorted[i], sorted[j]);
        const double med = sorted.empty() ? 0.0 : sorted[sorted.size() / 2];
        printf("{\"program\": \"gs4d_sweep\", \"n_gpus\": %d, \"splats\": %zu, \"frames\": %d, \"width\": %d, \"height\": %d, \"frames_per_gather_per_rank\": %d, \"frame_lanes\": %d, "
               "\"sweeps\": %d, \"ms_per_sweep\": %.4f, \"ms_per_frame\": %.5f, \"splats_per_s\": %.6g, \"frames_crc32\": \"%08x\", \"unordered_draws\": %llu, \"keygen_in_draw\": %llu, "
               "\"aborted_discarded\": %llu, \"batch_buffers\": %d, \"comm_stream_busy_ms_per_rank\": [",
               world, n, a.frames, a.width, a.height, G, lanes, a.sweeps, med * 1e3, med * 1e3 / a.frames, med > 0 ? (double)n * a.frames / med : 0.0,
               a.verify ? crc : 0u, (unsigned long long)(st[7] & 0xFFFFFFFFu), (unsigned long long)(st[6] >> 32), (unsigned long long)(st[2] >> 32), NB);
        for (int r = 0; r < world; ++r) printf("%s%.3f", r ? ", " : "", comm_all[2 * r]);
        printf("], \"comm_stream_tail_ms_per_rank\": [");
        for (int r = 0; r < world; ++r) printf("%s%.3f", r ? ", " : "", comm_all[2 * r + 1]);
        printf("], \"rank0_frames_pct_of_equal_share\": %d, \"frames_per_rank\": [", pct);
        for (int r = 0; r < world; ++r) printf("%s%zu", r ? ", " : "", frames_of[r].size());
        printf("], \"gather_format\": \"RGBA8\"}\n");
        fflush(stdout);
    }
    gs4d_destroy(ctx);
    for (int x = 0; x < NB; ++x) { if (rank != 0) (void)hipFree(batch[x]); (void)hipFree(gathered[x]); (void)hipEventDestroy(ev_free[x]); }
    for (int i = 0; i < max_gathers; ++i) { (void)hipEventDestroy(g0[i]); (void)hipEventDestroy(g1[i]); }
    (void)hipFree(dmax);
    (void)hipStreamDestroy(stream);
    ncclCommDestroy(comm);
    if (rank == 0) unlink(idfile.c_str());
    return 0;
}

// BASELINE.json configs[4]'s shape: ONE frame per step; splats cannot be sharded (the blend order is global), tiles can.  Every rank generates
// the keys of and sorts ALL splats, builds the tile lists of and composites only the rows of 8x8 tiles ty % world == rank
// (gs4d_set_tile_shard), packs its rows (gs4d_read_band_rgba8_device) and sends the band to rank 0, which holds the image as `world`
// bands.  --frames is the number of steps per timed window here; the frame is rendered at t = t_max / 2.
int sweep_tiles(const Args& a, int rank, int world, gs4d_ctx* ctx, ncclComm_t comm, hipStream_t stream, gs4d_buf data, const std::vector<gs4d_buf>& keys, const std::vector<gs4d_buf>& idx, int lanes, const float cam_pos[3]) {
    const size_t n = a.splats;
    GSOK(gs4d_set_tile_shard(ctx, rank, world));
    const int tiles_y = (a.height + 7) / 8;
    std::vector<size_t> band_bytes(world, 0), band_off(world + 1, 0);
    for (int ty = 0; ty < tiles_y; ++ty) band_bytes[ty % world] += (size_t)std::min(8, a.height - ty * 8) * a.width * 4;
    for (int r = 0; r < world; ++r) band_off[r + 1] = band_off[r] + band_bytes[r];
    int rows = 0;
    GSOK(gs4d_band_rows(ctx, &rows));
    if ((size_t)rows * a.width * 4 != band_bytes[rank]) { fprintf(stderr, "[rank %d] band size mismatch\n", rank); return 1; }
    uint8_t* band[2] = { nullptr, nullptr }; uint8_t* image[2] = { nullptr, nullptr }; double* dmax = nullptr;
    for (int x = 0; x < 2; ++x) {
        if (rank == 0) { HIPOK(hipMalloc(&image[x], band_off[world])); band[x] = image[x]; }
        else HIPOK(hipMalloc(&band[x], std::max<size_t>(band_bytes[rank], 4)));
    }
    HIPOK(hipMalloc(&dmax, sizeof(double)));
    const float t = a.t_max * 0.5f;
    GSOK(gs4d_set_uniform_1f(ctx, GS4D_U_TIME, t));
    uint64_t step_no = 0;
    auto step = [&]() -> int {
        const int b = (int)(step_no % (uint64_t)lanes), x = (int)(step_no & 1u);
        ++step_no;
        GSOK(gs4d_clear(ctx));
        GSOK(gs4d_keygen(ctx, data, t, cam_pos, keys[b], idx[b], n, GS4D_KEY_REF_INV_EUCLID));
        GSOK(gs4d_sort_pairs(ctx, keys[b], idx[b], n));
        GSOK(gs4d_bind_storage(ctx, 1, idx[b]));
        GSOK(gs4d_draw_instanced(ctx, n));
        if (band_bytes[rank]) GSOK(gs4d_read_band_rgba8_device(ctx, band[x], band_bytes[rank]));
        NCCLOK(ncclGroupStart());
        if (rank == 0) { for (int r = 1; r < world; ++r) if (band_bytes[r]) NCCLOK(ncclRecv(image[x] + band_off[r], band_bytes[r], ncclUint8, r, comm, stream)); }
        else if (band_bytes[rank]) NCCLOK(ncclSend(band[x], band_bytes[rank], ncclUint8, 0, comm, stream));
        NCCLOK(ncclGroupEnd());
        return 0;
    };
    auto fence = [&]() -> int {
        GSOK(gs4d_finish(ctx));
        HIPOK(hipStreamSynchronize(stream));
        HIPOK(hipMemset(dmax, 0, sizeof(double)));
        NCCLOK(ncclAllReduce(dmax, dmax, 1, ncclDouble, ncclMax, comm, stream));
        HIPOK(hipStreamSynchronize(stream));
        return 0;
    };
    uint32_t crc = 0u;
    if (a.verify) {
        if (step() || fence()) return 1;
        if (rank == 0) {
            // bands -> image rows: tile row ty lives in rank ty % world's band, the rank's tile rows in ascending ty
            std::vector<uint8_t> bands(band_off[world]), img((size_t)a.width * a.height * 4);
            HIPOK(hipMemcpy(bands.data(), image[(step_no - 1) & 1u], bands.size(), hipMemcpyDeviceToHost));
            std::vector<size_t> cur(band_off.begin(), band_off.end() - 1);
            for (int ty = 0; ty < tiles_y; ++ty) {
                const size_t bytes = (size_t)std::min(8, a.height - ty * 8) * a.width * 4;
                memcpy(img.data() + (size_t)ty * 8 * a.width * 4, bands.data() + cur[ty % world], bytes);
                cur[ty % world] += bytes;
            }
            crc = crc32_update(0u, img.data(), img.size());
            if (!a.dump.empty()) { mkdir(a.dump.c_str(), 0755); if (!write_file(a.dump + "/frame_tiles.rgba8", img.data(), img.size())) return 1; }
        }
    }
    for (int w = 0; w < a.warmup * 4; ++w) { if (step()) return 1; }
    if (fence()) return 1;
    std::vector<double> secs;
    for (int s = 0; s < a.sweeps; ++s) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < a.frames; ++k) { if (step()) return 1; }
        if (fence()) return 1;
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        HIPOK(hipMemcpy(dmax, &el, sizeof el, hipMemcpyHostToDevice));
        NCCLOK(ncclAllReduce(dmax, dmax, 1, ncclDouble, ncclMax, comm, stream));
        HIPOK(hipStreamSynchronize(stream));
        HIPOK(hipMemcpy(&el, dmax, sizeof el, hipMemcpyDeviceToHost));
        secs.push_back(el);
    }
    uint64_t st[8];
    GSOK(gs4d_get_stats(ctx, st));
    if (rank == 0) {
        std::vector<double> sorted = secs;
        for (size_t i = 0; i < sorted.size(); ++i) for (size_t j = i + 1; j < sorted.size(); ++j) if (sorted[j] < sorted[i]) std::swap(sorted[i], sorted[j]);
        const double med = sorted.empty() ? 0.0 : sorted[sorted.size() / 2];
        printf("{\"program\": \"gs4d_sweep\", \"mode\": \"shard_tiles\", \"n_gpus\": %d, \"splats\": %zu, \"steps_per_window\": %d, \"width\": %d, \"height\": %d, \"frame_lanes\": %d, "
               "\"windows\": %d, \"ms_per_frame\": %.5f, \"splats_per_s\": %.6g, \"image_crc32\": \"%08x\", \"tile_list_entries_rank0\": %llu, \"unordered_draws\": %llu}\n",
               world, n, a.frames, a.width, a.height, lanes, a.sweeps, med * 1e3 / a.frames, med > 0 ? (double)n * a.frames / med : 0.0, crc,
               (unsigned long long)(st[0] & 0xFFFFFFFFull), (unsigned long long)(st[7] & 0xFFFFFFFFu));
        fflush(stdout);
    }
    for (int x = 0; x < 2; ++x) { if (rank != 0) (void)hipFree(band[x]); (void)hipFree(image[x]); }
    (void)hipFree(dmax);
    return 0;
}

} // namespace

int main(int argc, char** argv) {
    g_started = (long)time(nullptr);
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", k.c_str()); exit(2); } return argv[++i]; };
        if (k == "--gpus") a.gpus = atoi(val());
        else if (k == "--splats") a.splats = (size_t)atoll(val());
        else if (k == "--frames") a.frames = atoi(val());
        else if (k == "--gather-every") a.gather_every = atoi(val());
        else if (k == "--fake-comm-us") a.fake_comm_us = atoi(val());
        else if (k == "--comm-priority") a.comm_priority = atoi(val());
        else if (k == "--batch-buffers") a.batch_buffers = std::min(8, std::max(2, atoi(val())));
        else if (k == "--rank0-frames-pct") a.rank0_pct = std::min(100, std::max(0, atoi(val())));
        else if (k == "--sweeps") a.sweeps = atoi(val());
        else if (k == "--warmup") a.warmup = atoi(val());
        else if (k == "--width") a.width = atoi(val());
        else if (k == "--height") a.height = atoi(val());
        else if (k == "--dump") a.dump = val();
        else if (k == "--png") a.png = val();
        else if (k == "--png-every") a.png_every = atoi(val());
        else if (k == "--no-verify") a.verify = false;
        else if (k == "--shard-tiles") a.shard_tiles = true;
        else { fprintf(stderr, "usage: gs4d_sweep [--gpus N] [--splats n] [--frames 256] [--gather-every 8] [--sweeps 3] [--warmup 1] [--width W --height H] [--dump dir] [--png prefix [--png-every 32]] [--no-verify] [--shard-tiles] [--rank0-frames-pct P] [--batch-buffers 2] [--comm-priority 0|1|2]\n"); return 2; }
    }
    if (a.gpus < 1 || a.frames < 1 || a.splats < 1 || a.png_every < 1) { fprintf(stderr, "bad arguments\n"); return 2; }
    const char* er = getenv("RANK"); const char* ew = getenv("WORLD_SIZE");
    if (er && ew) {                                                         // one rank of a job somebody else launched
        const int rank = atoi(er), world = atoi(ew), local = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
        const std::string port = getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0";
        const std::string idfile = std::string("/tmp/gs4d_sweep_") + port + ".id";
        // the job's nonce: the launcher's run id where it sets one (torchrun: TORCHELASTIC_RUN_ID; GS4D_SWEEP_JOB to name it yourself); without
        // one a rank accepts only a file written after it started itself (rank 0 deletes a leftover before it writes)
        std::string nonce;
        if (const char* j = getenv("GS4D_SWEEP_JOB")) nonce = j;
        else if (const char* j2 = getenv("TORCHELASTIC_RUN_ID")) nonce = j2;
        return run_rank(a, rank, world, local, idfile, nonce);
    }
    const std::string idfile = "/tmp/gs4d_sweep_" + std::to_string((long)getpid()) + ".id";
    unlink(idfile.c_str());
    const std::string nonce = "pid" + std::to_string((long)getpid());
    if (a.gpus == 1) return run_rank(a, 0, 1, 0, idfile, nonce);
    // one process per GPU, forked here — before this process has made a single HIP call
    std::vector<pid_t> kids;
    for (int r = 0; r < a.gpus; ++r) {
        const pid_t p = fork();
        if (p < 0) { perror("fork"); return 1; }
        if (p == 0) _exit(run_rank(a, r, a.gpus, r, idfile, nonce));
        kids.push_back(p);
    }
    int rc = 0;
    for (pid_t p : kids) { int stt = 0; if (waitpid(p, &stt, 0) < 0 || !WIFEXITED(stt) || WEXITSTATUS(stt) != 0) rc = 1; }
    unlink(idfile.c_str());
    return rc;
}
