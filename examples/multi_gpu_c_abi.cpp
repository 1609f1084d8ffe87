// Multi-GPU use of the C ABI from ONE process (the reference's own process model: one C++ program):
// one dsp_handle per GPU, one host thread per handle, objects block-sharded over the GPUs with no data-path collective, and ONE
// RCCL gather of the per-object results (dsp_gather_batch_results: device -> ncclGather over xGMI -> host, once).
//
//   python examples/export_example_data.py /tmp/dsp_example 16        # decoder.bin + objects.bin (fixture decoder, 16 synthetic objects)
//   g++ -std=c++17 -O2 -Iinclude examples/multi_gpu_c_abi.cpp -o /tmp/multi_gpu_c_abi -Ldsp_slam_amd/lib -ldspgn -Wl,-rpath,$PWD/dsp_slam_amd/lib -pthread
//   /tmp/multi_gpu_c_abi /tmp/dsp_example
//
// Prints one line per GPU and the gathered result's checksum; exits non-zero unless the sharded result equals, bit for bit, the result
// of all objects on one GPU (objects are independent: tests/test_gpu_configs.py::test_shard_equals_unsharded).  Works with one GPU too
// (a communicator of one rank), which is how the GPU tests run it.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "dsp_gn.h"

namespace {

struct Decoder {
    int32_t n_layers = 0, code_len = 0, latent_in = -1;
    std::vector<int32_t> out_dims, in_dims;
    std::vector<std::vector<float>> w, b;
};

struct Objects {
    int32_t n = 0;
    std::vector<int64_t> pts_off{0}, ray_off{0}, depth_off{0};
    std::vector<float> pts, rays, depth, t;
};

bool read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes; }

bool load_decoder(const std::string& path, Decoder& d) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    int32_t head[3];
    bool ok = read_exact(f, head, sizeof head);
    d.n_layers = head[0]; d.code_len = head[1]; d.latent_in = head[2];
    for (int k = 0; ok && k < d.n_layers; ++k) {
        int32_t dims[2];
        ok = read_exact(f, dims, sizeof dims);
        d.out_dims.push_back(dims[0]); d.in_dims.push_back(dims[1]);
        d.w.emplace_back((size_t)dims[0] * dims[1]);
        d.b.emplace_back((size_t)dims[0]);
        ok = ok && read_exact(f, d.w.back().data(), d.w.back().size() * 4) && read_exact(f, d.b.back().data(), d.b.back().size() * 4);
    }
    fclose(f);
    return ok;
}

bool load_objects(const std::string& path, Objects& o) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = read_exact(f, &o.n, 4);
    for (int i = 0; ok && i < o.n; ++i) {
        int32_t sz[3];      // surface points, rays, foreground depths
        ok = read_exact(f, sz, sizeof sz);
        const size_t p0 = o.pts.size(), r0 = o.rays.size(), d0 = o.depth.size(), t0 = o.t.size();
        o.t.resize(t0 + 16); o.pts.resize(p0 + (size_t)sz[0] * 3); o.rays.resize(r0 + (size_t)sz[1] * 3); o.depth.resize(d0 + sz[2]);
        ok = ok && read_exact(f, o.t.data() + t0, 64) && read_exact(f, o.pts.data() + p0, (size_t)sz[0] * 12) &&
             read_exact(f, o.rays.data() + r0, (size_t)sz[1] * 12) && read_exact(f, o.depth.data() + d0, (size_t)sz[2] * 4);
        o.pts_off.push_back(o.pts_off.back() + sz[0]);
        o.ray_off.push_back(o.ray_off.back() + sz[1]);
        o.depth_off.push_back(o.depth_off.back() + sz[2]);
    }
    fclose(f);
    return ok;
}

// objects [a, b) as a device-resident batch on handle h
dsp_batch* make_batch(dsp_handle* h, const dsp_gn_params& prm, const Objects& o, int a, int b) {
    std::vector<int64_t> po, ro, dof;
    for (int i = a; i <= b; ++i) {
        po.push_back(o.pts_off[i] - o.pts_off[a]);
        ro.push_back(o.ray_off[i] - o.ray_off[a]);
        dof.push_back(o.depth_off[i] - o.depth_off[a]);
    }
    dsp_batch* bt = nullptr;
    const int rc = dsp_batch_create(h, &prm, b - a, po.data(), o.pts.data() + 3 * o.pts_off[a], ro.data(), o.rays.data() + 3 * o.ray_off[a], dof.data(),
                                    o.depth.data() + o.depth_off[a], o.t.data() + 16 * (size_t)a, nullptr, &bt);
    if (rc != DSP_OK) { fprintf(stderr, "dsp_batch_create: %s\n", dsp_last_error(h)); return nullptr; }
    return bt;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <dir with decoder.bin and objects.bin> [max GPUs]\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    Decoder dec;
    Objects obj;
    if (!load_decoder(dir + "/decoder.bin", dec) || !load_objects(dir + "/objects.bin", obj) || obj.n < 1) { fprintf(stderr, "cannot read example data in %s\n", dir.c_str()); return 2; }
    int n_gpu = dsp_device_count();
    if (argc > 2) n_gpu = std::min(n_gpu, atoi(argv[2]));
    if (n_gpu < 1) { fprintf(stderr, "no gfx950 device\n"); return 2; }
    n_gpu = std::min(n_gpu, (int)obj.n);

    std::vector<const float*> wp, bp;
    for (int k = 0; k < dec.n_layers; ++k) { wp.push_back(dec.w[k].data()); bp.push_back(dec.b[k].data()); }
    dsp_decoder_desc desc{dec.n_layers, dec.code_len, dec.latent_in, dec.out_dims.data(), dec.in_dims.data(), wp.data(), bp.data()};
    dsp_gn_params prm{1.f, 100.f, 0.25f, 1e7f, 0.2f, 0.025f, 1.f, 1.f, 10, 50, 0.01f, 5};      // configs/config_kitti.json of the reference

    std::vector<dsp_handle*> handles(n_gpu, nullptr);
    for (int g = 0; g < n_gpu; ++g)
        if (dsp_create(&desc, g, &handles[g]) != DSP_OK) { fprintf(stderr, "dsp_create(device %d): %s\n", g, dsp_last_error(nullptr)); return 1; }

    // contiguous, deliberately UNEVEN blocks of objects per GPU (GPU g takes a share ~ g + 1, at least one object: the gather pads uneven
    // blocks to the largest); one host thread per handle runs its batch
    std::vector<int> bound(n_gpu + 1, 0);
    {
        const int64_t tri = (int64_t)n_gpu * (n_gpu + 1) / 2;
        for (int g = 1; g <= n_gpu; ++g) {
            int b = (int)((int64_t)obj.n * ((int64_t)g * (g + 1) / 2) / tri);
            b = std::max(b, bound[g - 1] + 1);                       // never an empty shard
            bound[g] = std::min(b, (int)obj.n - (n_gpu - g));        // ... and room for one object on every later GPU
        }
        bound[n_gpu] = obj.n;
    }
    std::vector<dsp_batch*> batches(n_gpu, nullptr);
    std::vector<int> rc(n_gpu, 0);
    std::vector<std::thread> threads;
    for (int g = 0; g < n_gpu; ++g)
        threads.emplace_back([&, g] {
            const int a = bound[g], b = bound[g + 1];
            batches[g] = make_batch(handles[g], prm, obj, a, b);
            rc[g] = batches[g] ? dsp_batch_run(batches[g]) : DSP_E_ARG;
            dsp_stats st;
            if (rc[g] == DSP_OK && dsp_batch_stats(batches[g], &st) == DSP_OK)
                printf("GPU %d: objects [%d, %d), %.1f ms on the device, prepass guard trips %.0f\n", g, a, b, st.ms_total, st.prepass_guard_trips);
        });
    for (auto& t : threads) t.join();
    for (int g = 0; g < n_gpu; ++g)
        if (rc[g] != DSP_OK) { fprintf(stderr, "GPU %d failed: %s\n", g, dsp_last_error(handles[g])); return 1; }

    std::vector<float> gathered((size_t)obj.n * DSP_RESULT_WIDTH);
    if (dsp_gather_batch_results(batches.data(), n_gpu, gathered.data()) != DSP_OK) { fprintf(stderr, "gather: %s\n", dsp_last_error(handles[0])); return 1; }

    // the same objects on ONE GPU: objects are independent, so the rows must be the same bits
    dsp_batch* all = make_batch(handles[0], prm, obj, 0, obj.n);
    if (!all || dsp_batch_run(all) != DSP_OK) { fprintf(stderr, "single-GPU run: %s\n", dsp_last_error(handles[0])); return 1; }
    std::vector<float> t((size_t)obj.n * 16), c((size_t)obj.n * DSP_CODE_LEN), l(obj.n), single((size_t)obj.n * DSP_RESULT_WIDTH);
    std::vector<int32_t> s(obj.n);
    dsp_batch_results(all, t.data(), c.data(), l.data(), s.data());
    dsp_pack_results(obj.n, t.data(), c.data(), l.data(), s.data(), single.data());
    const bool same = memcmp(single.data(), gathered.data(), single.size() * 4) == 0;
    uint32_t sum = 0;
    int good = 0;
    for (size_t i = 0; i < gathered.size(); ++i) { uint32_t u; memcpy(&u, &gathered[i], 4); sum = sum * 31u + u; }
    for (int i = 0; i < obj.n; ++i) good += gathered[(size_t)i * DSP_RESULT_WIDTH + 81] == 0.f;
    printf("%d objects on %d GPU(s): %d good, gathered checksum %08x, sharded == single GPU: %s\n", obj.n, n_gpu, good, sum, same ? "yes" : "NO");

    dsp_batch_destroy(all);
    for (int g = 0; g < n_gpu; ++g) { dsp_batch_destroy(batches[g]); dsp_destroy(handles[g]); }
    return same ? 0 : 1;
}
