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
// Rendezvous: rank 0 writes the ncclUniqueId to a file under /tmp (renamed into place), the others wait for it.
// Presentation is software-pipelined as a swap chain is (frame j is queued first, then frame j-1 — the previous image — is packed to
// RGBA8 into the batch) and every G presented frames the batch goes to rank 0: ncclSend / ncclRecv in one group, on a stream of this
// program's that the context knows as the caller's stream (gs4d_set_stream): a pack waits for the gather that still reads its slot, a
// gather for the packs it sends — by events, inside the library.  Every rank presents ceil(256 / N) times, so all ranks make the same
// collective calls whatever the world size.
#include "../../include/gs4d.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

namespace {

struct Args {
    int gpus = 1, frames = 256, gather_every = 8, sweeps = 3, warmup = 1, width = 1920, height = 1080;
    size_t splats = 1000000;
    float t_max = 50.0f;
    std::string dump;            // directory: rank 0 writes records.bin and frame_####.rgba8 of the verification sweep (tests)
    std::string png;             // prefix: rank 0 writes <prefix>####.png for every --png-every-th frame of the verification sweep
    int png_every = 32;
    bool verify = true;          // one untimed sweep whose frames are copied to the host on rank 0 and check-summed in frame order
};

#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLOK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, ncclGetErrorString(r_)); return 1; } } while (0)
#define GSOK(call) do { int r_ = (call); if (r_ != GS4D_OK) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, gs4d_last_error(ctx)); return 1; } } while (0)
int g_rank = 0;

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

int run_rank(const Args& a, int rank, int world, int local_rank, const std::string& idfile) {
    g_rank = rank;
    gs4d_ctx* ctx = nullptr;
    HIPOK(hipSetDevice(local_rank));
    // ---- communicator ----
    ncclUniqueId id;
    if (rank == 0) {
        NCCLOK(ncclGetUniqueId(&id));
        const std::string tmp = idfile + ".tmp";
        if (!write_file(tmp, &id, sizeof id) || rename(tmp.c_str(), idfile.c_str()) != 0) { fprintf(stderr, "cannot write %s\n", idfile.c_str()); return 1; }
    } else {
        bool got = false;
        for (int tries = 0; tries < 6000 && !got; ++tries) {                  // up to 60 s
            FILE* f = fopen(idfile.c_str(), "rb");
            if (f) { got = fread(&id, 1, sizeof id, f) == sizeof id; fclose(f); }
            if (!got) usleep(10000);
        }
        if (!got) { fprintf(stderr, "[rank %d] no rendezvous file %s\n", rank, idfile.c_str()); return 1; }
    }
    ncclComm_t comm;
    NCCLOK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream;
    HIPOK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    // ---- scene, resident on the device ----
    const size_t n = a.splats;
    std::vector<float> rec;
    make_records(n, rec);
    if (gs4d_create(local_rank, a.width, a.height, &ctx) != GS4D_OK) { fprintf(stderr, "[rank %d] gs4d_create: %s\n", rank, gs4d_last_error(nullptr)); return 1; }
    GSOK(gs4d_set_stream(ctx, stream));
    uint64_t st[8];
    GSOK(gs4d_get_stats(ctx, st));
    const int lanes = (int)(st[6] & 0xFFFFFFFFu);
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

    // ---- frames of this rank, batches ----
    std::vector<int> mine;
    for (int k = rank; k < a.frames; k += world) mine.push_back(k);
    const int most = (a.frames + world - 1) / world;                        // presentations per rank and sweep (rank 0 has the most frames)
    const int G = a.gather_every < 1 ? 1 : a.gather_every;
    const size_t fbytes = (size_t)a.width * a.height * 4;
    uint8_t* batch = nullptr; uint8_t* gathered = nullptr; double* dmax = nullptr;
    if (rank == 0) { HIPOK(hipMalloc(&gathered, (size_t)world * G * fbytes)); batch = gathered; }      // rank 0 packs straight into its slice of the gathered batch
    else HIPOK(hipMalloc(&batch, G * fbytes));
    HIPOK(hipMemset(batch, 0, G * fbytes));
    HIPOK(hipMalloc(&dmax, sizeof(double)));
    HIPOK(hipMemset(dmax, 0, sizeof(double)));
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
    auto gather = [&](int batch_no, bool verify) -> int {
        NCCLOK(ncclGroupStart());
        if (rank == 0) { for (int r = 1; r < world; ++r) NCCLOK(ncclRecv(gathered + (size_t)r * G * fbytes, G * fbytes, ncclUint8, r, comm, stream)); }
        else NCCLOK(ncclSend(batch, G * fbytes, ncclUint8, 0, comm, stream));
        NCCLOK(ncclGroupEnd());
        if (verify && rank == 0) {
            host_frames.resize((size_t)world * G * fbytes);
            HIPOK(hipMemcpyAsync(host_frames.data(), gathered, host_frames.size(), hipMemcpyDeviceToHost, stream));
            HIPOK(hipStreamSynchronize(stream));
            for (int r = 0; r < world; ++r)
                for (int p = 0; p < G; ++p) {
                    const int k = r + (batch_no * G + p) * world;            // slot p of rank r's batch holds its frame batch_no * G + p
                    if (batch_no * G + p >= most || k >= a.frames) continue;
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
        int presented = 0;
        auto present = [&](int j, int frames_back) -> int {
            if (j < (int)mine.size()) GSOK(gs4d_read_frame_rgba8_device(ctx, frames_back, batch + (size_t)(presented % G) * fbytes, fbytes));
            ++presented;
            if (presented % G == 0 || presented == most) { if (gather((presented - 1) / G, verify)) return 1; }
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
    auto fence = [&]() -> int {                                             // everything queued is done on every rank
        GSOK(gs4d_finish(ctx));
        HIPOK(hipStreamSynchronize(stream));
        NCCLOK(ncclAllReduce(dmax, dmax, 1, ncclDouble, ncclMax, comm, stream));
        HIPOK(hipStreamSynchronize(stream));
        return 0;
    };

    if (rank == 0 && !a.dump.empty()) { mkdir(a.dump.c_str(), 0755); if (!write_file(a.dump + "/records.bin", rec.data(), rec.size() * 4)) return 1; }
    if (a.verify) { if (sweep(true) || fence()) return 1; }
    for (int w = 0; w < a.warmup; ++w) { if (sweep(false)) return 1; }
    if (fence()) return 1;
    std::vector<double> secs;
    for (int s = 0; s < a.sweeps; ++s) {
        const auto t0 = std::chrono::steady_clock::now();
        if (sweep(false) || fence()) return 1;
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        HIPOK(hipMemcpy(dmax, &el, sizeof el, hipMemcpyHostToDevice));     // maximum over ranks
        NCCLOK(ncclAllReduce(dmax, dmax, 1, ncclDouble, ncclMax, comm, stream));
        HIPOK(hipStreamSynchronize(stream));
        HIPOK(hipMemcpy(&el, dmax, sizeof el, hipMemcpyDeviceToHost));
        secs.push_back(el);
    }
    GSOK(gs4d_get_stats(ctx, st));
    if (rank == 0) {
        uint32_t crc = 0u;
        for (int k = 0; k < a.frames; ++k) crc = crc32_update(crc, (const uint8_t*)&frame_crc[k], 4);
        std::vector<double> sorted = secs;
        for (size_t i = 0; i < sorted.size(); ++i) for (size_t j = i + 1; j < sorted.size(); ++j) if (sorted[j] < sorted[i]) std::swap(sorted[i], sorted[j]);
        const double med = sorted.empty() ? 0.0 : sorted[sorted.size() / 2];
        printf("{\"program\": \"gs4d_sweep\", \"n_gpus\": %d, \"splats\": %zu, \"frames\": %d, \"width\": %d, \"height\": %d, \"frames_per_gather_per_rank\": %d, \"frame_lanes\": %d, "
               "\"sweeps\": %d, \"ms_per_sweep\": %.4f, \"ms_per_frame\": %.5f, \"splats_per_s\": %.6g, \"frames_crc32\": \"%08x\", \"unordered_draws\": %llu, \"keygen_in_draw\": %llu}\n",
               world, n, a.frames, a.width, a.height, G, lanes, a.sweeps, med * 1e3, med * 1e3 / a.frames, med > 0 ? (double)n * a.frames / med : 0.0,
               a.verify ? crc : 0u, (unsigned long long)(st[7] & 0xFFFFFFFFu), (unsigned long long)(st[6] >> 32));
        fflush(stdout);
    }
    gs4d_destroy(ctx);
    if (rank != 0) (void)hipFree(batch);
    (void)hipFree(gathered); (void)hipFree(dmax);
    (void)hipStreamDestroy(stream);
    ncclCommDestroy(comm);
    if (rank == 0) unlink(idfile.c_str());
    return 0;
}

} // namespace

int main(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", k.c_str()); exit(2); } return argv[++i]; };
        if (k == "--gpus") a.gpus = atoi(val());
        else if (k == "--splats") a.splats = (size_t)atoll(val());
        else if (k == "--frames") a.frames = atoi(val());
        else if (k == "--gather-every") a.gather_every = atoi(val());
        else if (k == "--sweeps") a.sweeps = atoi(val());
        else if (k == "--warmup") a.warmup = atoi(val());
        else if (k == "--width") a.width = atoi(val());
        else if (k == "--height") a.height = atoi(val());
        else if (k == "--dump") a.dump = val();
        else if (k == "--png") a.png = val();
        else if (k == "--png-every") a.png_every = atoi(val());
        else if (k == "--no-verify") a.verify = false;
        else { fprintf(stderr, "usage: gs4d_sweep [--gpus N] [--splats n] [--frames 256] [--gather-every 8] [--sweeps 3] [--warmup 1] [--width W --height H] [--dump dir] [--png prefix [--png-every 32]] [--no-verify]\n"); return 2; }
    }
    if (a.gpus < 1 || a.frames < 1 || a.splats < 1 || a.png_every < 1) { fprintf(stderr, "bad arguments\n"); return 2; }
    const char* er = getenv("RANK"); const char* ew = getenv("WORLD_SIZE");
    if (er && ew) {                                                         // one rank of a job somebody else launched
        const int rank = atoi(er), world = atoi(ew), local = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
        const std::string idfile = std::string("/tmp/gs4d_sweep_") + (getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0") + ".id";
        return run_rank(a, rank, world, local, idfile);
    }
    const std::string idfile = "/tmp/gs4d_sweep_" + std::to_string((long)getpid()) + ".id";
    unlink(idfile.c_str());
    if (a.gpus == 1) return run_rank(a, 0, 1, 0, idfile);
    // one process per GPU, forked here — before this process has made a single HIP call
    std::vector<pid_t> kids;
    for (int r = 0; r < a.gpus; ++r) {
        const pid_t p = fork();
        if (p < 0) { perror("fork"); return 1; }
        if (p == 0) _exit(run_rank(a, r, a.gpus, r, idfile));
        kids.push_back(p);
    }
    int rc = 0;
    for (pid_t p : kids) { int stt = 0; if (waitpid(p, &stt, 0) < 0 || !WIFEXITED(stt) || WEXITSTATUS(stt) != 0) rc = 1; }
    unlink(idfile.c_str());
    return rc;
}
