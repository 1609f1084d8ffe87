// Host side of libdspgn: decoder packing, device memory, the per-iteration launch sequence and the C ABI
// declared in include/dsp_gn.h.  No Python / PyTorch dependency.
#include "dsp_gn.h"
#include "dsp_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is dlopen()ed by dsp_gather_results, never linked

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <mutex>
#include <string>
#include <vector>

using namespace dsp;

namespace {

thread_local std::string g_create_error;

// Prepass margins are calibrated per decoder at dsp_create (calibrate_prepass): delta = 5 x the largest |sdf_lp - sdf_fp32| over
// 64 k unit-ball points x 4 codes, not below these floors (what the cars fixture needs: measured 1.05e-4 f16, 6.8e-4 bf16 over the
// 38 M audited samples of the bench workload, profiles/parity_r02.md).
constexpr float PREPASS_DELTA_F16 = 5e-4f, PREPASS_DELTA_BF16 = 3e-3f;

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            char buf_[512];                                                                                 \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw std::runtime_error(buf_);                                                                 \
        }                                                                                                   \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        free();
        if (count == 0) count = 1;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count) { if (count > n) alloc(count + count / 4); }
    void free() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    ~DevBuf() { free(); }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

struct dsp_handle {
    std::mutex mu;             // every entry point that touches the handle's stream / scratch holds it (see guarded())
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int n_cu = 256;
    // packed decoder
    DevBuf<float> wstream, wsplit, bias_tab, codew, b0, blat;
    int split_off[4] = {0, 0, 0, 0}, split_len[4] = {0, 0, 0, 0}, split_len_fwd[4] = {0, 0, 0, 0};   // per-wave layout of wsplit (latency-form kernels)
    std::vector<float> h_codew, h_b0, h_blat;   // host copies for the single-shot decoder calls
    float b_last = 0.f;
    int wlast_row = 0, w0_row = 0;
    int n_bias_rows = 0, n_fwd = 0, n_pass_all = 0, chunks_fwd = 0, chunks_all = 0;
    int lat_tile = 27, code_len = CODE_LEN;
    PassDesc pass[MAX_PASSES];
    // low-precision prepass (mlp_lp_kernel.hip): packed 16-bit weight streams [0] = f16, [1] = bf16 and their pass table
    DevBuf<uint16_t> wlp[2];
    LpPass lp_pass[LP_MAX_PASSES];
    int lp_n_pass = 0, lp_chunks = 0;
    bool lp_ok = false;        // decoder geometry supported by the prepass kernel (hidden width 512)
    float lp_err[2] = {0.f, 0.f};      // calibration at dsp_create: largest |sdf_lp - sdf_fp32| over the calibration points, per dtype
    float lp_delta[2] = {0.f, 0.f};    // the margin derived from it (prepass_delta default)
    // scratch of the single-shot decoder calls
    DevBuf<float4> s_pts;
    DevBuf<float> s_code, s_out, s_cbias;
    DevBuf<int4> s_tiles;
    DevBuf<int> s_ntiles;
    DevBuf<unsigned long long> s_clk;
    // mesh extraction: case table, scan scratch and the last extracted mesh (device resident until fetched)
    DevBuf<McTables> mc_tab;
    DevBuf<int2> mc_blocks;
    DevBuf<long long> mc_totals;
    DevBuf<float> mc_vol, mc_verts;
    DevBuf<int> mc_vidmap, mc_faces;
    int64_t mesh_nv = -1, mesh_nf = -1;
};

// ------------------------------------------------------------------------------------------------
// decoder packing: weights -> chunk stream in MFMA operand order (see mlp_kernel.hip / DESIGN.md)
// ------------------------------------------------------------------------------------------------
namespace {

struct NetView {
    int n_layers, hidden, lat;
    int code_len = CODE_LEN, in_dim = IN_DIM;   // of the decoder (code_len 32 or 64)
    int width = WIDTH;                           // hidden width of the decoder (<= 512: narrower nets are embedded with zero rows / columns)
    std::vector<int> out_dims, in_dims;
    std::vector<const float*> w, b;
    // layer k: input-slab row -> original weight column (or -1).  The code columns of the latent_in layer are not part
    // of the forward stream (they are a per-object bias, k_code_bias) but their gradient rows 448..511 are produced by the
    // backward pass; layer 0 never goes through the forward stream at all and only its 64 code rows through the backward one.
    int lat_row0() const { return WIDTH - in_dim; }
    std::vector<int> colmap(int k, bool backward) const {
        std::vector<int> m(WIDTH, -1);
        if (k == 0) {
            if (backward) for (int r = 0; r < code_len; ++r) m[r] = r;
        } else if (k == lat) {
            const int p = lat_row0();                                        // slab row of x: 445 (code_len 64) or 477 (32)
            const int hr = width - in_dim;                                   // rows of the previous layer's output
            for (int r = 0; r < hr; ++r) m[r] = r;
            for (int r = 0; r < 3; ++r) m[p + r] = hr + code_len + r;        // xyz re-injected at slab rows p .. p+2
            if (backward) for (int r = 0; r < code_len; ++r) m[p + 3 + r] = hr + r;   // code gradient rows p+3 ..
        } else {
            for (int r = 0; r < in_dims[k]; ++r) m[r] = r;
        }
        return m;
    }
};

template <class F>
void pack_pass(std::vector<float>& stream, int nog, int nchunks, F value) {
    const size_t base = stream.size();
    stream.resize(base + (size_t)nog * nchunks * (CHUNK_BYTES / 4));
    float* dst = stream.data() + base;
    for (int og = 0; og < nog; ++og)
        for (int c = 0; c < nchunks; ++c) {
            float* ch = dst + ((size_t)og * nchunks + c) * (CHUNK_BYTES / 4);
            for (int sl = 0; sl < KSTEPS_PER_CHUNK; ++sl)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int s = KSTEPS_PER_CHUNK * c + sl;
                        const int krow = 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3);
                        const int orow = 64 * og + 16 * j + (lane & 15);
                        ch[(sl * 64 + lane) * 4 + j] = value(orow, krow);
                    }
        }
}

struct PackedNet {
    std::vector<float> stream, bias;
    std::vector<float> codew;   // [2][512][64]: W0[:, :64] and W_lat[:, code columns], row-major
    std::vector<float> b0, blat;
    float b_last = 0.f;
    int wlast_row = 0, w0_row = 0;
    int n_bias_rows = 0, n_fwd = 0, n_pass_all = 0, chunks_fwd = 0, chunks_all = 0;
    int lat_tile = 27, code_len = CODE_LEN;   // 16-row slab tile holding the re-injected xyz (27: rows 445..447, 29: 477..479)
    PassDesc pass[MAX_PASSES];
};

void pack_decoder_host(PackedNet* h, const dsp_decoder_desc* d) {
    NetView nv;
    nv.n_layers = d->n_layers;
    nv.hidden = d->n_layers - 1;
    nv.lat = d->latent_in;
    if (d->code_len != 64 && d->code_len != 32) throw std::invalid_argument("code_len must be 64 or 32");
    if (nv.hidden < 2 || nv.hidden > 8) throw std::invalid_argument("need 2..8 hidden layers");
    if (nv.lat < 2 || nv.lat >= nv.hidden) throw std::invalid_argument("latent_in must name one hidden layer >= 2");
    nv.code_len = d->code_len;
    nv.in_dim = d->code_len + 3;
    nv.width = d->out_dims[0];
    // Narrower decoders run embedded in the 512-row slabs (zero rows / columns: an fmaf with a zero weight leaves the accumulator
    // unchanged, so the results are those of a kernel built for the narrow width).  The re-injected input of the latent_in layer
    // sits at fixed slab rows (445.. or 477..), so the previous layer's output must end below them.
    if (nv.width < 16 || nv.width > WIDTH || nv.width % 16) throw std::invalid_argument("hidden width must be a multiple of 16, at most 512");
    if (nv.width - nv.in_dim > nv.lat_row0()) throw std::invalid_argument("hidden width too large for this code length");
    for (int k = 0; k < d->n_layers; ++k) {
        nv.out_dims.push_back(d->out_dims[k]);
        nv.in_dims.push_back(d->in_dims[k]);
        nv.w.push_back(d->weights[k]);
        nv.b.push_back(d->biases[k]);
        const int want_in = (k == 0) ? nv.in_dim : nv.width;
        const int want_out = (k == nv.hidden) ? 1 : (k + 1 == nv.lat ? nv.width - nv.in_dim : nv.width);
        if (d->in_dims[k] != want_in || d->out_dims[k] != want_out)
            throw std::invalid_argument("unsupported decoder geometry (one hidden width, input code_len + 3, output 1, one latent_in layer)");
    }
    h->lat_tile = nv.lat_row0() / 16;
    h->code_len = nv.code_len;
    std::vector<float>& stream = h->stream;
    std::vector<float>& bias = h->bias;
    stream.clear();
    bias.assign((size_t)(nv.hidden + 3) * WIDTH, 0.f);
    memset(h->pass, 0, sizeof h->pass);
    int np = 0, chunk = 0;
    // forward passes (layer 0 is evaluated on the VALU from the per-object code bias + the xyz columns)
    for (int k = 1; k < nv.hidden; ++k) {
        const std::vector<int> cm = nv.colmap(k, false);
        const int od = nv.out_dims[k], id = nv.in_dims[k];
        const float* W = nv.w[k];
        PassDesc& p = h->pass[np++];
        p.nog = (int16_t)((std::max(od, 1) + 63) / 64);
        p.nchunks = (int16_t)(k == nv.lat ? (nv.lat_row0() + 3 + 63) / 64 : 8);          // latent_in: K = 445 + 3 rows (7 chunks) or 477 + 3 (8)
        p.bias_row = (int16_t)(k == nv.lat ? -2 : k - 1);    // -2: per-object bias (code contribution + b_lat)
        p.relu = 1;
        p.mask_slot = (int16_t)k;
        p.kind = (int16_t)(k == nv.lat ? 2 : 1);
        p.chunk_base = chunk;
        pack_pass(stream, p.nog, p.nchunks, [&](int orow, int krow) -> float {
            if (orow >= od || krow >= WIDTH || cm[krow] < 0) return 0.f;
            return W[(size_t)orow * id + cm[krow]];
        });
        chunk += p.nog * p.nchunks;
        if (k != nv.lat) for (int o = 0; o < od; ++o) bias[(size_t)(k - 1) * WIDTH + o] = nv.b[k][o];
    }
    h->n_fwd = np;
    h->chunks_fwd = chunk;
    h->wlast_row = nv.hidden - 1;
    h->w0_row = nv.hidden;
    for (int o = 0; o < nv.width; ++o) bias[(size_t)h->wlast_row * WIDTH + o] = nv.w[nv.hidden][o];
    for (int c3 = 0; c3 < 3; ++c3)
        for (int o = 0; o < nv.width; ++o) bias[(size_t)(h->w0_row + c3) * WIDTH + o] = nv.w[0][(size_t)o * nv.in_dim + nv.code_len + c3];
    h->b_last = nv.b[nv.hidden][0];
    h->n_bias_rows = nv.hidden + 3;
    // code columns of layer 0 and of the latent_in layer, for the per-object bias
    h->codew.assign((size_t)2 * WIDTH * CODE_LEN, 0.f);
    h->b0.assign(WIDTH, 0.f);
    h->blat.assign(WIDTH, 0.f);
    std::copy(nv.b[0], nv.b[0] + nv.width, h->b0.begin());
    std::copy(nv.b[nv.lat], nv.b[nv.lat] + nv.width, h->blat.begin());
    for (int o = 0; o < nv.width; ++o)
        for (int c = 0; c < nv.code_len; ++c) {       // code columns beyond code_len stay zero: the optimiser carries 64-entry codes
            h->codew[(size_t)o * CODE_LEN + c] = nv.w[0][(size_t)o * nv.in_dim + c];
            h->codew[(size_t)(WIDTH + o) * CODE_LEN + c] = nv.w[nv.lat][(size_t)o * nv.width + (nv.width - nv.in_dim) + c];
        }
    // backward passes (transposed weights): output rows = the forward layer's INPUT slab rows
    for (int k = nv.hidden - 1; k >= 0; --k) {
        const std::vector<int> cm = nv.colmap(k, true);
        const int od = nv.out_dims[k], id = nv.in_dims[k];
        const float* W = nv.w[k];
        PassDesc& p = h->pass[np++];
        p.nog = (int16_t)(k == 0 ? 1 : 8);                   // layer 0: only the 64 code rows (xyz on the VALU)
        p.nchunks = (int16_t)((od + 63) / 64);
        p.bias_row = -1;
        p.relu = 0;
        p.mask_slot = (int16_t)(k - 1);
        p.kind = (int16_t)(k == nv.lat ? 4 : (k == 0 ? 5 : 3));
        p.chunk_base = chunk;
        pack_pass(stream, p.nog, p.nchunks, [&](int orow, int krow) -> float {
            if (krow >= od || orow >= WIDTH || cm[orow] < 0) return 0.f;
            return W[(size_t)krow * id + cm[orow]];
        });
        chunk += p.nog * p.nchunks;
    }
    h->n_pass_all = np;
    h->chunks_all = chunk;
}

// ---- low-precision prepass stream (mlp_lp_kernel.hip) --------------------------------------------------------------------
uint16_t lp_bits(float v, bool bf) {      // fp32 -> f16 / bf16 bits, round to nearest even (= v_cvt_pk_{f16,bf16}_f32)
    uint16_t r;
    if (bf) {
        uint32_t u;
        memcpy(&u, &v, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        r = (uint16_t)(u >> 16);
    } else {
        const _Float16 h = (_Float16)v;
        memcpy(&r, &h, 2);
    }
    return r;
}
float lp_value(uint16_t b, bool bf) {
    if (bf) { const uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}
// part p (1..3) of the split w = w1 + w2 + w3 (each a 16-bit value)
float lp_part(float w, int p, bool bf) {
    const float w1 = lp_value(lp_bits(w, bf), bf);
    if (p == 1) return w1;
    const float w2 = lp_value(lp_bits(w - w1, bf), bf);
    if (p == 2) return w2;
    return lp_value(lp_bits(w - w1 - w2, bf), bf);
}

struct PackedLp {
    std::vector<uint16_t> stream;
    LpPass pass[LP_MAX_PASSES];
    int n_pass = 0, chunks = 0;
};

// Returns false when the decoder's geometry is outside what mlp_lp_kernel<*, 4> is built for (the prepass is then off and
// every sample goes through the fp32 kernel, as in round 1).
bool pack_decoder_lp_host(PackedLp* out, const dsp_decoder_desc* d, bool bf) {
    const int hidden = d->n_layers - 1, lat = d->latent_in;
    constexpr int NCH = 4, NOG = 2 * NCH;
    if (hidden < 2 || hidden > LP_MAX_PASSES || lat < 2 || lat >= hidden) return false;
    if (hidden & 1) return false;      // the kernel's last-layer body reads slab X: an even number of passes (DeepSDF's 8)
    const int in_dim = d->code_len + 3, width = d->out_dims[0];            // narrower nets are embedded with zero rows / columns
    if (width < 16 || width > WIDTH) return false;
    for (int k = 0; k < d->n_layers; ++k) {
        const int want_in = k == 0 ? in_dim : width;
        const int want_out = k == hidden ? 1 : (k + 1 == lat ? width - in_dim : width);
        if (d->in_dims[k] != want_in || d->out_dims[k] != want_out) return false;
    }
    const int lat_rows = width - in_dim;                                  // slab rows of the latent_in layer (445 / 477 for width 512)
    const int lat_ksteps = (lat_rows + 15) / 16;
    const int xyz0 = LP_KSTEPS_PER_CHUNK * NCH - LP_XYZ_KSTEPS;           // first xyz k-step of the latent_in layer
    if (lat_ksteps > xyz0) return false;
    const int tsel = bf ? 1 : 0;
    out->stream.clear();
    memset(out->pass, 0, sizeof out->pass);
    int chunk = 0;
    for (int k = 0; k < hidden; ++k) {
        LpPass& p = out->pass[k];
        const int od = d->out_dims[k], id = d->in_dims[k];
        const float* W = d->weights[k];
        p.kind = (int16_t)(k == 0 ? 0 : (k == lat ? 2 : 1));
        p.nog = NOG;               // layers with fewer than 512 outputs are padded with zero rows: every pass is the same straight-line code
        p.nchunks = (int16_t)(k == 0 ? 1 : NCH);
        p.bias_row = (int16_t)(k == 0 ? -3 : (k == lat ? -2 : k - 1));
        p.npad = (int16_t)(k == lat ? std::min(3, xyz0 - lat_ksteps) : 0);   // (a narrower net's unused slab rows are zero anyway: every layer writes them)
        p.last = (int16_t)(k == hidden - 1);
        p.chunk_base = chunk;
        const int slab_rows = k == 0 ? 0 : (k == lat ? lat_rows : id);
        const int xyz_col = k == 0 ? d->code_len : lat_rows + d->code_len;   // first of the three xyz columns of W
        const int xyz_first = k == 0 ? 0 : (k == lat ? xyz0 : 1 << 20);       // k-step of the first xyz operand
        const size_t base = out->stream.size();
        out->stream.resize(base + (size_t)p.nog * p.nchunks * (CHUNK_BYTES / 2), 0);
        uint16_t* dst = out->stream.data() + base;
        for (int g = 0; g < p.nog; ++g)
            for (int c = 0; c < p.nchunks; ++c)
                for (int sl = 0; sl < LP_KSTEPS_PER_CHUNK; ++sl)
                    for (int j = 0; j < 2; ++j)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int s = LP_KSTEPS_PER_CHUNK * c + sl, hh = lane >> 5;
                                const int orow = 64 * g + 32 * j + (lane & 31);
                                float v = 0.f;
                                if (orow < od) {
                                    if (s >= xyz_first && s < xyz_first + LP_XYZ_KSTEPS) {
                                        const int kk = 8 * hh + e, t = kk / 3;
                                        const int ent = t < 5 ? LP_XYZ_TERMS[tsel][s - xyz_first][t] : 0;
                                        if (ent) v = lp_part(W[(size_t)orow * id + xyz_col + kk % 3], ent & 3, bf);
                                    } else if (s < xyz_first) {
                                        const int krow = 16 * s + 8 * (e >> 2) + 4 * hh + (e & 3);
                                        if (krow < slab_rows) v = W[(size_t)orow * id + krow];
                                    }
                                }
                                dst[((((size_t)(g * p.nchunks + c) * LP_KSTEPS_PER_CHUNK + sl) * 2 + j) * 64 + lane) * 8 + e] = lp_bits(v, bf);
                            }
        chunk += p.nog * p.nchunks;
    }
    out->n_pass = hidden;
    out->chunks = chunk;
    return true;
}

// Latency form (mlp_split_kernel): wave w of a workgroup produces output groups 2w and 2w+1 of every pass and streams only
// their chunks.  Returns the chunk ids of the throughput stream in the order the four waves consume them (wave 0's whole
// stream, then wave 1's, ...; inside a wave: pass, own group, chunk -- forward passes first, then backward), and each wave's
// offset / length / forward-prefix length in chunks.
std::vector<int> split_chunk_order(const PackedNet& pn, int off[4], int len[4], int len_fwd[4]) {
    std::vector<int> order;
    for (int w = 0; w < 4; ++w) {
        off[w] = (int)order.size();
        len_fwd[w] = 0;
        for (int ps = 0; ps < pn.n_pass_all; ++ps) {
            if (ps == pn.n_fwd) len_fwd[w] = (int)order.size() - off[w];
            for (int ol = 0; ol < 2; ++ol) {
                const int og = 2 * w + ol;
                if (og >= pn.pass[ps].nog) continue;
                for (int c = 0; c < pn.pass[ps].nchunks; ++c) order.push_back(pn.pass[ps].chunk_base + og * pn.pass[ps].nchunks + c);
            }
        }
        len[w] = (int)order.size() - off[w];
    }
    return order;
}

void pack_decoder(dsp_handle* h, const dsp_decoder_desc* d) {
    PackedNet pn;
    pack_decoder_host(&pn, d);
    h->b_last = pn.b_last; h->n_bias_rows = pn.n_bias_rows; h->n_fwd = pn.n_fwd; h->n_pass_all = pn.n_pass_all;
    h->chunks_fwd = pn.chunks_fwd; h->chunks_all = pn.chunks_all;
    h->wlast_row = pn.wlast_row; h->w0_row = pn.w0_row;
    h->lat_tile = pn.lat_tile; h->code_len = pn.code_len;
    h->h_codew = pn.codew; h->h_b0 = pn.b0; h->h_blat = pn.blat;
    h->codew.alloc(pn.codew.size());
    HIP_TRY(hipMemcpy(h->codew.p, pn.codew.data(), pn.codew.size() * 4, hipMemcpyHostToDevice));
    h->b0.alloc(WIDTH); h->blat.alloc(WIDTH);
    HIP_TRY(hipMemcpy(h->b0.p, pn.b0.data(), WIDTH * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->blat.p, pn.blat.data(), WIDTH * 4, hipMemcpyHostToDevice));
    memcpy(h->pass, pn.pass, sizeof h->pass);
    h->wstream.alloc(pn.stream.size());
    HIP_TRY(hipMemcpy(h->wstream.p, pn.stream.data(), pn.stream.size() * 4, hipMemcpyHostToDevice));
    {   // latency form: per-wave copy of the stream (split_chunk_order)
        const size_t cf = CHUNK_BYTES / 4;
        const std::vector<int> order = split_chunk_order(pn, h->split_off, h->split_len, h->split_len_fwd);
        std::vector<float> split;
        split.reserve(pn.stream.size());
        for (int chunk : order) split.insert(split.end(), pn.stream.begin() + (size_t)chunk * cf, pn.stream.begin() + (size_t)(chunk + 1) * cf);
        if (split.size() != pn.stream.size()) throw std::logic_error("split weight stream does not cover the stream");
        h->wsplit.alloc(split.size());
        HIP_TRY(hipMemcpy(h->wsplit.p, split.data(), split.size() * 4, hipMemcpyHostToDevice));
    }
    h->bias_tab.alloc(pn.bias.size());
    HIP_TRY(hipMemcpy(h->bias_tab.p, pn.bias.data(), pn.bias.size() * 4, hipMemcpyHostToDevice));
    h->lp_ok = true;
    for (int bf = 0; bf < 2 && h->lp_ok; ++bf) {
        PackedLp pl;
        if (!pack_decoder_lp_host(&pl, d, bf != 0)) { h->lp_ok = false; break; }
        h->wlp[bf].alloc(pl.stream.size());
        HIP_TRY(hipMemcpy(h->wlp[bf].p, pl.stream.data(), pl.stream.size() * 2, hipMemcpyHostToDevice));
        memcpy(h->lp_pass, pl.pass, sizeof h->lp_pass);
        h->lp_n_pass = pl.n_pass;
        h->lp_chunks = pl.chunks;
    }
}

LpArgs make_lp_args(const dsp_handle* h, bool bf) {
    LpArgs a;
    memset(&a, 0, sizeof a);
    a.wstream = h->wlp[bf ? 1 : 0].p;
    a.bias_tab = h->bias_tab.p;
    a.b_last = h->b_last;
    a.n_bias_rows = h->n_bias_rows;
    a.wlast_row = h->wlast_row;
    a.n_pass = h->lp_n_pass;
    a.total_chunks = h->lp_chunks;
    memcpy(a.pass, h->lp_pass, sizeof a.pass);
    return a;
}

// per-object code bias on the host (single-shot calls); same arithmetic as k_code_bias
void code_bias_host(const std::vector<float>& codew, const std::vector<float>& b0, const std::vector<float>& blat, const float* code,
                    float* out /*1024*/) {
    for (int which = 0; which < 2; ++which)
        for (int o = 0; o < WIDTH; ++o) {
            float acc = which == 0 ? b0[o] : blat[o];
            const float* w = codew.data() + ((size_t)which * WIDTH + o) * CODE_LEN;
            for (int c = 0; c < CODE_LEN; ++c) acc = fmaf(w[c], code[c], acc);
            out[which * WIDTH + o] = acc;
        }
}

MlpArgs make_mlp_args(const dsp_handle* h, int mode) {   // mode: 0/1 forward, 2 forward+backward, 3 backward only (mlp_kernel)
    MlpArgs a;
    memset(&a, 0, sizeof a);
    a.bias_tab = h->bias_tab.p;
    a.b_last = h->b_last;
    a.n_bias_rows = h->n_bias_rows;
    a.wlast_row = h->wlast_row;
    a.w0_row = h->w0_row;
    a.seed_slot = h->n_fwd;        // mask slot of the last hidden layer (slot = layer index, layer 0 has slot 0)
    a.lat_tile = h->lat_tile;
    a.wsplit = h->wsplit.p;
    memcpy(a.split_off, h->split_off, sizeof a.split_off);
    memcpy(a.split_len, h->split_len, sizeof a.split_len);
    memcpy(a.split_len_fwd, h->split_len_fwd, sizeof a.split_len_fwd);
    if (mode == 3) {               // only the backward half of the stream and of the pass table
        const int n_bwd = h->n_pass_all - h->n_fwd;
        a.wstream = h->wstream.p + (size_t)h->chunks_fwd * (CHUNK_BYTES / 4);
        a.n_fwd = 0;
        a.n_pass = n_bwd;
        a.total_chunks = h->chunks_all - h->chunks_fwd;
        memcpy(a.pass, h->pass + h->n_fwd, n_bwd * sizeof(PassDesc));
    } else {
        a.wstream = h->wstream.p;
        a.n_fwd = h->n_fwd;
        a.n_pass = mode == 2 ? h->n_pass_all : h->n_fwd;
        a.total_chunks = mode == 2 ? h->chunks_all : h->chunks_fwd;
        memcpy(a.pass, h->pass, sizeof a.pass);
    }
    return a;
}

// n_codes objects x the same n points, already in h->s_pts (device): decoder forward or forward+gradient into h->s_out,
// output row = code * n + point.  Asynchronous on h->stream.
// lp: 0 = fp32 kernels, 1 = f16 prepass kernel, 2 = bf16 prepass kernel (forward only)
void decode_resident_points(dsp_handle* h, const float* codes, int64_t n_codes, int64_t n, bool bwd, int lp = 0) {
    if (lp && (bwd || !h->lp_ok)) throw std::invalid_argument("low-precision decode: forward only, hidden width 512");
    const int tile_pts = lp ? LP_TILE_PTS : TILE_PTS;
    const int64_t ntile = (n + tile_pts - 1) / tile_pts;
    if (ntile * n_codes > (int64_t)1 << 30 || n * n_codes > (int64_t)1 << 31) throw std::invalid_argument("decode request too large");
    const int nt = (int)(ntile * n_codes);
    std::vector<int4> tiles(nt);
    for (int64_t c = 0; c < n_codes; ++c)
        for (int64_t i = 0; i < ntile; ++i)
            tiles[c * ntile + i] = make_int4((int)(i * tile_pts), (int)std::min<int64_t>(tile_pts, n - i * tile_pts), (int)c, (int)(c * n));
    const size_t n_out = (size_t)n * n_codes;
    h->s_code.ensure((size_t)CODE_LEN * n_codes);
    h->s_cbias.ensure((size_t)2 * WIDTH * n_codes);
    std::vector<float> cb((size_t)2 * WIDTH * n_codes);
    for (int64_t c = 0; c < n_codes; ++c) code_bias_host(h->h_codew, h->h_b0, h->h_blat, codes + c * CODE_LEN, cb.data() + c * 2 * WIDTH);
    HIP_TRY(hipMemcpyAsync(h->s_cbias.p, cb.data(), cb.size() * 4, hipMemcpyHostToDevice, h->stream));
    h->s_tiles.ensure(nt);
    h->s_ntiles.ensure(1);
    h->s_out.ensure(bwd ? n_out * GRAD_STRIDE : n_out);
    HIP_TRY(hipMemcpyAsync(h->s_code.p, codes, (size_t)CODE_LEN * n_codes * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->s_tiles.p, tiles.data(), nt * sizeof(int4), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->s_ntiles.p, &nt, 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));     // the staging vectors above go out of scope
    h->s_clk.ensure(4);
    if (lp) {
        LpArgs la = make_lp_args(h, lp == 2);
        la.n_tiles = h->s_ntiles.p;
        la.tiles = h->s_tiles.p;
        la.pts = h->s_pts.p;
        la.code_bias = h->s_cbias.p;
        la.code_bias_stride = 2 * WIDTH;
        la.out_sdf = h->s_out.p;
        la.clk = h->s_clk.p;
        HIP_TRY(launch_mlp_lp(lp == 2, la, std::min(nt, h->n_cu), h->stream));
        return;
    }
    MlpArgs a = make_mlp_args(h, bwd ? 2 : 0);
    a.n_tiles = h->s_ntiles.p;
    a.tiles = h->s_tiles.p;
    a.pts = h->s_pts.p;
    a.codes = h->s_code.p;
    a.code_stride = CODE_LEN;
    a.code_bias = h->s_cbias.p;
    a.code_bias_stride = 2 * WIDTH;
    a.out_sdf = h->s_out.p;
    a.out_grad = h->s_out.p;
    h->s_clk.ensure(4);
    a.clk = h->s_clk.p;
    HIP_TRY(launch_mlp(bwd ? 2 : 0, a, std::min(nt, h->n_cu), h->stream));
}

// n_codes objects x the same n host points (object frame): forward or forward+gradient.  Output row = code * n + point.
void run_decoder_points(dsp_handle* h, const float* codes, int64_t n_codes, const float* pts, int64_t n, bool bwd, float* sdf_out,
                        float* grad_out, int lp = 0) {
    if (n <= 0 || n_codes <= 0) return;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<float4> p4((size_t)n);
    for (int64_t i = 0; i < n; ++i) p4[i] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], 0.f);
    h->s_pts.ensure(n);
    HIP_TRY(hipMemcpyAsync(h->s_pts.p, p4.data(), n * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    decode_resident_points(h, codes, n_codes, n, bwd, lp);
    const size_t n_out = (size_t)n * n_codes;
    if (!bwd) {
        HIP_TRY(hipMemcpyAsync(sdf_out, h->s_out.p, n_out * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    } else {
        std::vector<float> tmp(n_out * GRAD_STRIDE);
        HIP_TRY(hipMemcpyAsync(tmp.data(), h->s_out.p, tmp.size() * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < n_out; ++i) {
            if (grad_out) memcpy(grad_out + i * DSP_GRAD_DIM, tmp.data() + i * GRAD_STRIDE, DSP_GRAD_DIM * 4);
            if (sdf_out) sdf_out[i] = tmp[i * GRAD_STRIDE + 67];
        }
    }
}

// Prepass margin of THIS decoder: decode seeded unit-ball points with a few codes through the fp32 kernel and through both
// low-precision kernels and take 5 x the largest difference (never below the floors above).  A decoder whose 16-bit error were larger
// than the fixture's -- bigger activations, say -- gets a wider band instead of misclassified samples; dsp_prepass_calibration reports it.
void calibrate_prepass(dsp_handle* h) {
    if (!h->lp_ok) return;
    constexpr int N = 16384, NCODE = 4;
    std::vector<float> pts((size_t)N * 3), codes((size_t)NCODE * CODE_LEN, 0.f), ref((size_t)N * NCODE), lp((size_t)N * NCODE);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xFFFFFF) / 16777216.f; };
    for (int i = 0; i < N;) {                      // uniform in the unit ball (the optimiser only decodes in-sphere samples)
        const float x = 2.f * rnd() - 1.f, y = 2.f * rnd() - 1.f, z = 2.f * rnd() - 1.f;
        if (x * x + y * y + z * z >= 1.f) continue;
        pts[3 * i] = x; pts[3 * i + 1] = y; pts[3 * i + 2] = z;
        ++i;
    }
    for (int c = 1; c < NCODE; ++c)                // code 0 = the zero start; the others ~ +-0.15 uniform on the decoder's own entries
        for (int k = 0; k < h->code_len; ++k) codes[(size_t)c * CODE_LEN + k] = 0.3f * (rnd() - 0.5f);
    run_decoder_points(h, codes.data(), NCODE, pts.data(), N, false, ref.data(), nullptr, 0);
    for (int bf = 0; bf < 2; ++bf) {
        run_decoder_points(h, codes.data(), NCODE, pts.data(), N, false, lp.data(), nullptr, bf ? DSP_PREPASS_BF16 : DSP_PREPASS_F16);
        float worst = 0.f;
        for (size_t i = 0; i < ref.size(); ++i) {
            const float d = std::fabs(lp[i] - ref[i]);
            if (!(d <= worst)) worst = std::isfinite(d) ? d : 1.f;      // a non-finite prepass value: make the band cover everything
        }
        h->lp_err[bf] = worst;
        h->lp_delta[bf] = std::min(0.5f, std::max(5.f * worst, bf ? PREPASS_DELTA_BF16 : PREPASS_DELTA_F16));
    }
}

// marching cubes over a device-resident volume; the mesh stays in h->mc_verts / mc_faces until dsp_mesh_fetch
void extract_mesh_device(dsp_handle* h, const float* vol, int n0, int n1, int n2, float level, float spacing, float origin) {
    if ((int64_t)n0 * n1 * n2 > ((int64_t)1 << 30)) throw std::invalid_argument("volume too large");
    const int n_pts = n0 * n1 * n2;
    if (!h->mc_tab.p) {
        McTables t;
        mc_build_tables(t);
        h->mc_tab.alloc(1);
        HIP_TRY(hipMemcpy(h->mc_tab.p, &t, sizeof t, hipMemcpyHostToDevice));
    }
    h->mc_blocks.ensure(mc_num_blocks(n_pts));
    h->mc_totals.ensure(2);
    h->mesh_nv = h->mesh_nf = -1;
    HIP_TRY(launch_mc_count(vol, n0, n1, n2, level, h->mc_tab.p, h->mc_blocks.p, h->mc_totals.p, h->stream));
    long long tot[2];
    HIP_TRY(hipMemcpyAsync(tot, h->mc_totals.p, sizeof tot, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (tot[0] > ((long long)1 << 30) || tot[1] > ((long long)1 << 30)) throw std::invalid_argument("mesh too large");
    h->mc_verts.ensure((size_t)std::max<long long>(tot[0], 1) * 3);
    h->mc_faces.ensure((size_t)std::max<long long>(tot[1], 1) * 3);
    h->mc_vidmap.ensure((size_t)n_pts * 3);
    if (tot[0] > 0)
        HIP_TRY(launch_mc_emit(vol, n0, n1, n2, level, h->mc_tab.p, h->mc_blocks.p, spacing, origin, h->mc_verts.p, h->mc_vidmap.p,
                               h->mc_faces.p, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->mesh_nv = tot[0];
    h->mesh_nf = tot[1];
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// batches
// ------------------------------------------------------------------------------------------------
struct dsp_batch {
    dsp_handle* h = nullptr;
    dsp_gn_params prm;
    int B = 0, D = 0, maxR = 0, maxM = 0, n_slices = 1;
    bool pose_only = false, trace_on = false;
    int64_t sum_pts = 0, sum_rays = 0, sum_depth = 0, cap_s = 0, cap_j = 0;
    std::vector<ObjConst> oc_host;
    DevBuf<ObjConst> oc;
    DevBuf<ObjState> st;
    DevBuf<float> pts, rays, depth, t_in, codes_in, scale_in;
    bool have_codes = false;
    DevBuf<unsigned long long> raymask;
    DevBuf<int> raycnt, rayoff, kcnt, koff, mcnt, pcnt, poff, plist;
    DevBuf<unsigned char> ray_alive, ray_hint, ray_plo;
    DevBuf<unsigned short> maskbuf;   // relu masks of band samples, 512 B per sample (written by the forward passes)
    int split_rows = -1;      // jacobian launch in the latency form (16-point tiles, rows split over waves): -1 auto, 0 off, 1 on
    int mask_reuse = -1;      // render rows backward-only from exported relu masks: -1 auto, 0 off, 1 on (dsp_batch_set_mask_reuse)
    int n_ray_passes = 0;     // front-to-back forward passes per iteration (0 = pick from the batch size)
    std::vector<int> pass_bounds;   // optional explicit depth-index boundaries (n_passes + 1 entries, 0 .. D)
    int hint_margin = 2, hint_step = 8;   // adaptive passes: pass 0 = [0, hint + margin), middle pass = next `step` indices
    int speculative = -1;         // band samples straight into the jacobian launch (latency path): -1 auto, 0 off, 1 on
    DevBuf<int> srow, jrow;       // speculative band rows: jgrad row of a sample / of a kept render row
    int fused_bookkeeping = -1;   // per-object fused bookkeeping kernels: -1 auto (latency-sized batches), 0 off, 1 on
    int prepass = -1;         // low-precision classification pass in front of the fp32 forward decoder: -1 auto, 0 off, 1 f16, 2 bf16
    float prepass_delta = -1.f;   // margin added to cut_off on both sides (< 0: the dtype's default)
    bool prepass_audit = false;   // also decode every sample in fp32 and compare (tests / calibration)
    DevBuf<float> saudit;
    DevBuf<unsigned> audit_out;
    DevBuf<float> ray_res, ssdf, sdeds, jgrad, partials, trace, out_t, out_code, out_loss, rows;
    DevBuf<float4> spts, jpts;
    DevBuf<float2> jaux;
    DevBuf<unsigned char> alive;
    DevBuf<int4> tiles_f, tiles_j;
    DevBuf<int> n_tiles, out_status;
    DevBuf<double> counters, gsum;
    DevBuf<float> cbias;
    std::vector<hipEvent_t> ev;   // pairs around every decoder launch + [run start, run end]
    std::vector<int> ev_kind;
    dsp_stats stats;
    int iters_run = 0;
    ~dsp_batch() { for (auto e : ev) (void)hipEventDestroy(e); }
};

namespace {

dsp_batch* batch_build(dsp_handle* h, const dsp_gn_params* prm, int B, const int64_t* pts_off, const float* pts,
                       const int64_t* ray_off, const float* rays, const int64_t* depth_off, const float* depth,
                       const float* t_in, const float* codes_in, const float* scale_in, bool pose_only) {
    if (B <= 0) throw std::invalid_argument("n_objects must be positive");
    if (!pose_only && (prm->num_depth_samples < 2 || prm->num_depth_samples > MAX_DEPTH_SAMPLES))
        throw std::invalid_argument("num_depth_samples must be in [2, 64]");
    HIP_TRY(hipSetDevice(h->device));
    std::unique_ptr<dsp_batch> b(new dsp_batch);
    b->h = h;
    b->prm = *prm;
    b->B = B;
    b->D = pose_only ? 2 : prm->num_depth_samples;
    b->pose_only = pose_only;
    b->oc_host.resize(B);
    int64_t cap_s = 0, cap_j = 0;
    for (int i = 0; i < B; ++i) {
        ObjConst& c = b->oc_host[i];
        memset(&c, 0, sizeof c);
        c.pts_off = (int)pts_off[i];
        c.n_pts = (int)(pts_off[i + 1] - pts_off[i]);
        if (!pose_only) {
            c.ray_off = (int)ray_off[i];
            c.n_rays = (int)(ray_off[i + 1] - ray_off[i]);
            c.depth_off = (int)depth_off[i];
            c.n_fg = (int)(depth_off[i + 1] - depth_off[i]);
            if (c.n_fg > c.n_rays) throw std::invalid_argument("more depths than rays");
            if (c.n_rays >= (1 << 25)) throw std::invalid_argument("too many rays per object");
        }
        const int scap = pose_only ? 0 : round_up(c.n_rays * b->D, TILE_PTS);
        c.samp_off = (int)cap_s;
        cap_s += scap;
        c.jsdf_off = (int)cap_j;
        cap_j += round_up(c.n_pts, TILE_PTS);
        c.jren_off = (int)cap_j;
        cap_j += scap;
        b->maxR = std::max(b->maxR, c.n_rays);
        b->maxM = std::max(b->maxM, c.n_pts);
        if (cap_j > (int64_t)1 << 30) throw std::invalid_argument("batch too large for 32-bit point indices");
    }
    b->cap_s = cap_s;
    b->cap_j = cap_j;
    b->sum_pts = pts_off[B];
    b->sum_rays = pose_only ? 0 : ray_off[B];
    b->sum_depth = pose_only ? 0 : depth_off[B];
    b->n_slices = 16;   // fixed: the Gram summation order (hence every bit of the result) must not depend on the batch
    b->oc.alloc(B);
    b->st.alloc(B);
    HIP_TRY(hipMemcpy(b->oc.p, b->oc_host.data(), B * sizeof(ObjConst), hipMemcpyHostToDevice));
    b->pts.alloc(b->sum_pts * 3);
    HIP_TRY(hipMemcpy(b->pts.p, pts, b->sum_pts * 12, hipMemcpyHostToDevice));
    b->t_in.alloc((size_t)B * 16);
    HIP_TRY(hipMemcpy(b->t_in.p, t_in, (size_t)B * 64, hipMemcpyHostToDevice));
    b->codes_in.alloc((size_t)B * CODE_LEN);
    b->have_codes = codes_in != nullptr;
    if (codes_in) HIP_TRY(hipMemcpy(b->codes_in.p, codes_in, (size_t)B * CODE_LEN * 4, hipMemcpyHostToDevice));
    b->scale_in.alloc(B);
    if (scale_in) HIP_TRY(hipMemcpy(b->scale_in.p, scale_in, (size_t)B * 4, hipMemcpyHostToDevice));
    if (!pose_only) {
        b->rays.alloc(b->sum_rays * 3);
        HIP_TRY(hipMemcpy(b->rays.p, rays, b->sum_rays * 12, hipMemcpyHostToDevice));
        b->depth.alloc(b->sum_depth);
        if (b->sum_depth) HIP_TRY(hipMemcpy(b->depth.p, depth, b->sum_depth * 4, hipMemcpyHostToDevice));
        b->raymask.alloc(b->sum_rays);
        b->raycnt.alloc(b->sum_rays); b->rayoff.alloc(b->sum_rays);
        b->kcnt.alloc(b->sum_rays); b->koff.alloc(b->sum_rays); b->mcnt.alloc(b->sum_rays);
        b->pcnt.alloc(b->sum_rays); b->poff.alloc(b->sum_rays); b->plist.alloc(cap_s); b->ray_alive.alloc(b->sum_rays);
        b->ray_hint.alloc(b->sum_rays); b->ray_plo.alloc(b->sum_rays);
        b->maskbuf.alloc((size_t)cap_s * 256);
        b->ray_res.alloc(b->sum_rays);
        b->spts.alloc(cap_s); b->ssdf.alloc(cap_s); b->sdeds.alloc(cap_s);
        b->tiles_f.alloc(cap_s / SPLIT_TILE_PTS + B);     // sized for the latency form's 16-point tiles
    } else {
        b->alive.alloc(cap_j);
    }
    b->jpts.alloc(cap_j); b->jaux.alloc(cap_j);
    b->jgrad.alloc((size_t)cap_j * GRAD_STRIDE);
    b->tiles_j.alloc(cap_j / SPLIT_TILE_PTS + 2 * B);     // sized for the latency form's 16-point tiles
    b->n_tiles.alloc(4);
    b->counters.alloc(8);
    b->audit_out.alloc(4);
    b->partials.alloc((size_t)B * 2 * b->n_slices * 72 * 72);
    b->gsum.alloc((size_t)B * 2 * 72 * 72);
    b->cbias.alloc((size_t)B * 2 * WIDTH);
    b->out_t.alloc((size_t)B * 16); b->out_code.alloc((size_t)B * CODE_LEN); b->out_loss.alloc(B); b->out_status.alloc(B);
    memset(&b->stats, 0, sizeof b->stats);
    return b.release();
}

GnParamsDev dev_params(const dsp_batch* b) {
    GnParamsDev p;
    p.k1 = b->prm.k1; p.k2 = b->prm.k2; p.k3 = b->prm.k3; p.k4 = b->prm.k4;
    p.b1 = b->prm.b1; p.b2 = b->prm.b2; p.lr = b->prm.lr; p.s_damp = b->prm.s_damp; p.cut_off = b->prm.cut_off;
    p.n_depth = b->D; p.pose_only = b->pose_only ? 1 : 0;
    p.code_len = b->h->code_len;
    return p;
}

hipEvent_t next_event(dsp_batch* b, size_t& cursor) {
    if (cursor == b->ev.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        b->ev.push_back(e);
    }
    return b->ev[cursor++];
}

bool use_mask_reuse(const dsp_batch* b) {
    if (b->pose_only) return false;
    if (b->mask_reuse >= 0) return b->mask_reuse != 0;
    // not for latency-sized batches: with one or two objects every jacobian tile fits a single round over the CUs, so a
    // second launch adds a round instead of saving a forward sweep (tools/gpu_reuse_probe.py, cfg2 objects:
    // 1 object 38.1 ms on vs 32.3 off; 4 objects 42.1 obj/s vs 40.1; 8: 49.6 vs 45.5; 32: 49.4 vs 44.9)
    return b->sum_pts / TILE_PTS >= b->h->n_cu / 4;
}

// Jacobian launch in the latency form (mlp_split_kernel: 16-point tiles, each layer's rows split over the four waves, a tile
// takes ~1/3 of a 64-point tile's time)?  Worth it only while the 16-point tiles still fit a round or two over the CUs.
bool use_speculative_band(const dsp_batch* b);
bool use_split_rows(const dsp_batch* b) {
    if (use_mask_reuse(b)) return false;               // the backward-only launch has no latency form
    if (b->split_rows >= 0) return b->split_rows != 0;
    const double rows = (double)b->sum_pts + (b->pose_only ? 0.0 : (use_speculative_band(b) ? 0.16 : 0.045) * (double)b->sum_rays * b->D);   // M + typical K (or band)
    const double n_cu = b->h->n_cu;
    const double rounds16 = std::ceil(rows / SPLIT_TILE_PTS / n_cu), rounds64 = std::ceil(rows / TILE_PTS / n_cu);
    return 0.34 * rounds16 <= 0.8 * rounds64;
}

// The same question for the forward launches over ray samples (per pass roughly a third of the in-sphere samples).
bool use_split_fwd(const dsp_batch* b) {
    if (b->pose_only || use_mask_reuse(b)) return false;   // the mask-exporting forward has no latency form
    if (b->split_rows >= 0) return b->split_rows != 0;
    const double pts = 0.2 * (double)b->cap_s, n_cu = b->h->n_cu;   // tools/gpu_split_probe.py: 4 real-size objects 14.8 vs 17.3 ms forward; 1-2 cfg2 objects: 64-point tiles win
    const double rounds16 = std::ceil(pts / SPLIT_TILE_PTS / n_cu), rounds64 = std::ceil(pts / TILE_PTS / n_cu);
    return 0.30 * rounds16 <= 0.8 * rounds64;
}

// Latency-sized batches run the per-ray bookkeeping in its fused per-object form (k_front_fused / k_band_fused / k_render_fused:
// 3 launches instead of 11 per iteration, same device functions, same bits).  Large batches keep one thread block per 256 rays.
bool use_fused_bookkeeping(const dsp_batch* b) {
    if (b->pose_only) return false;
    if (b->fused_bookkeeping >= 0) return b->fused_bookkeeping != 0;
    // one workgroup per object: pays while an object's rays fit a few passes of its 16 waves (a real detection has <= 450 rays;
    // a cfg2-sized object with 2500 rays is faster through the per-256-ray launches: 22.1 vs 20.7 ms)
    return b->B <= 16 && b->maxR <= 768;
}

// Latency path, prepass on: the samples the prepass could not classify go STRAIGHT into the jacobian launch (forward + backward,
// their sdf scattered back for the occupancy scan) instead of a forward launch of their own followed by forward + backward of the
// kept ones.  One decoder launch less per iteration; the backward sweep of the ~25 % band samples that are not kept afterwards is
// wasted, so only while surface points + band samples fit one round of 16-point tiles.  The Gram kernel reads each kept row's
// gradient where that launch left it (jrow), in the same row order: bit-identical results.
int prepass_mode(const dsp_batch* b);
bool use_speculative_band(const dsp_batch* b) {
    if (!use_fused_bookkeeping(b) || !prepass_mode(b) || use_mask_reuse(b)) return false;
    if (b->speculative >= 0) return b->speculative != 0;
    return ((double)b->sum_pts + 0.16 * (double)b->cap_s) / SPLIT_TILE_PTS <= 1.0 * b->h->n_cu;
}


int prepass_mode(const dsp_batch* b) {      // 0 off, 1 f16, 2 bf16
    if (b->pose_only || !b->h->lp_ok) return 0;
    if (b->prepass >= 0) return b->prepass;
    return DSP_PREPASS_F16;
}
float prepass_delta(const dsp_batch* b) {
    if (b->prepass_delta >= 0.f) return b->prepass_delta;
    return b->h->lp_delta[prepass_mode(b) == DSP_PREPASS_BF16 ? 1 : 0];     // calibrated for THIS decoder at dsp_create
}

// what: 0 = forward pass over the current sample list (with mask reuse: relu masks of band samples exported), 1 = jacobian
// launch, forward + backward (with mask reuse the surface points only, else surface points and render rows), 2 = jacobian
// of the kept render rows, backward only from the exported masks
// 3 = low-precision prepass over the current sample list, 4 = audit: fp32 forward over every in-sphere sample into saudit
void launch_decoder(dsp_batch* b, int what, size_t& cursor, bool whole_list = false) {
    dsp_handle* h = b->h;
    if (what == 3) {
        const int pm = prepass_mode(b);
        LpArgs la = make_lp_args(h, pm == DSP_PREPASS_BF16);
        la.n_tiles = b->n_tiles.p;
        la.tiles = b->tiles_f.p;
        la.pts = b->spts.p;
        la.index = whole_list ? nullptr : b->plist.p;     // whole_list: the tiles cover every in-sphere sample in place
        la.code_bias = b->cbias.p;
        la.code_bias_stride = 2 * WIDTH;
        la.out_sdf = b->ssdf.p;
        hipEvent_t e0 = next_event(b, cursor), e1 = next_event(b, cursor);
        b->ev_kind.push_back(2);
        HIP_TRY(hipEventRecord(e0, h->stream));
        HIP_TRY(launch_mlp_lp(pm == DSP_PREPASS_BF16, la, h->n_cu, h->stream));
        HIP_TRY(hipEventRecord(e1, h->stream));
        return;
    }
    if (what == 4) {
        MlpArgs a = make_mlp_args(h, 0);
        a.n_tiles = b->n_tiles.p + 3;
        a.tiles = b->tiles_j.p;                 // free at this point of the iteration (the jacobian list is built later)
        a.pts = b->spts.p;
        a.code_bias = b->cbias.p;
        a.code_bias_stride = 2 * WIDTH;
        a.out_sdf = b->saudit.p;
        HIP_TRY(launch_mlp(0, a, h->n_cu, h->stream));
        return;
    }
    const bool reuse = use_mask_reuse(b);
    const int mode = what == 0 ? (reuse ? 1 : 0) : (what == 1 ? 2 : 3);
    MlpArgs a = make_mlp_args(h, mode);
    if (what == 0) {
        a.n_tiles = b->n_tiles.p;
        a.tiles = b->tiles_f.p;
        a.pts = b->spts.p;
        a.index = b->plist.p;
    } else {
        a.tiles = b->tiles_j.p;
        a.pts = b->jpts.p;
        a.n_tiles = (what == 1 && reuse) ? b->n_tiles.p + 2 : b->n_tiles.p + 1;     // [2] = surface tiles, [1] = surface + render tiles
        a.tile_begin = what == 1 ? nullptr : b->n_tiles.p + 2;
    }
    a.codes = reinterpret_cast<const float*>(reinterpret_cast<const char*>(b->st.p) + offsetof(ObjState, code));
    a.code_stride = sizeof(ObjState) / 4;
    a.code_bias = b->cbias.p;
    a.code_bias_stride = 2 * WIDTH;
    a.out_sdf = b->ssdf.p;
    a.sdf_in = b->ssdf.p;
    a.mask_buf = b->maskbuf.p;
    a.th = b->prm.cut_off;
    a.out_grad = b->jgrad.p;
    if (what == 1 && use_speculative_band(b)) { a.sdf_scatter = b->ssdf.p; a.scatter_tile_begin = b->n_tiles.p + 2; }
    hipEvent_t e0 = next_event(b, cursor), e1 = next_event(b, cursor);
    b->ev_kind.push_back(what == 0 ? 0 : 1);
    HIP_TRY(hipEventRecord(e0, h->stream));
    if (what == 1 && use_split_rows(b))
        HIP_TRY(launch_mlp_split(true, a, h->n_cu, h->stream));
    else if (what == 0 && use_split_fwd(b))
        HIP_TRY(launch_mlp_split(false, a, h->n_cu, h->stream));
    else
        HIP_TRY(launch_mlp(mode, a, h->n_cu, h->stream));
    HIP_TRY(hipEventRecord(e1, h->stream));
}

void batch_code_bias(dsp_batch* b) {
    dsp_handle* h = b->h;
    const float* codes = reinterpret_cast<const float*>(reinterpret_cast<const char*>(b->st.p) + offsetof(ObjState, code));
    launch_code_bias(h->codew.p, h->b0.p, h->blat.p, codes, (int)(sizeof(ObjState) / 4), b->cbias.p, b->B, h->stream);
}

// the first half of an iteration up to and including the J rows' inputs (shared with the stand-alone terms)
void iteration_front(dsp_batch* b, size_t& cursor, bool do_render, bool code_bias = true) {
    dsp_handle* h = b->h;
    hipStream_t s = h->stream;
    const int B = b->B;
    if (code_bias) batch_code_bias(b);   // first iteration / stand-alone terms; later iterations get it from k_solve, which updated the code
    const bool fused = do_render && use_fused_bookkeeping(b);
    const bool spec = do_render && use_speculative_band(b);
    if (do_render) {
        if (fused) {
            launch_front_fused(b->oc.p, b->st.p, b->rays.p, b->pts.p, b->raymask.p, b->raycnt.p, b->rayoff.p, b->spts.p, b->ssdf.p, b->ray_alive.p,
                               b->jpts.p, b->jaux.p, b->D, B, s);
        } else {
            launch_sample_count(b->oc.p, b->st.p, b->rays.p, b->raymask.p, b->raycnt.p, b->D, b->maxR, B, s);
            launch_scan_rays(b->oc.p, b->st.p, b->raycnt.p, b->rayoff.p, 0, B, s);
            launch_sample_write(b->oc.p, b->st.p, b->rays.p, b->raymask.p, b->rayoff.p, b->spts.p, b->ssdf.p, b->ray_alive.p, b->D, b->maxR, B, s);
        }
        // Forward decoder, front to back with exact early ray termination (gn_kernels.hip, "front-to-back ray passes").
        //  * explicit pass count / boundaries (dsp_batch_set_ray_passes / _bounds): fixed depth-index ranges for all rays;
        //  * automatic (default): per-ray ranges steered by where each ray terminated in the previous GN iteration -- pass 0
        //    decodes [0, hint + 2), a middle pass the next 8 indices (only with enough tiles to fill the chip), the last
        //    pass the rest.  Fewer launches than fixed ranges and less overshoot behind the surface.
        // With the prepass on, these passes run the LOW-PRECISION kernel (a ray stops behind its first certainly-solid
        // sample), and one fp32 launch follows over the samples the prepass could not classify (k_band_count).
        const int pm = prepass_mode(b);
        const float thd = b->prm.cut_off + prepass_delta(b);
        std::vector<PassSpec> specs;
        const double tiles = 0.75 * (double)b->cap_s / (pm ? LP_TILE_PTS : TILE_PTS);     // expected forward tiles per iteration
        int fixed_passes = b->n_ray_passes;
        if (fixed_passes <= 0 && tiles >= 100.0 * h->n_cu) fixed_passes = 10;   // large batches: ten uniform ranges measured best
        // a prepass over every sample that fits a round and a half of 128-point tiles is one launch with no pass bookkeeping at all
        if (fixed_passes <= 0 && pm && (double)b->cap_s / LP_TILE_PTS <= 1.5 * h->n_cu) fixed_passes = 1;
        if (fixed_passes > 0) {
            const int n_passes = std::max(1, std::min(fixed_passes, b->D));
            std::vector<int> bounds = b->pass_bounds;
            if ((int)bounds.size() != n_passes + 1) {
                bounds.assign(n_passes + 1, 0);
                for (int p = 0; p <= n_passes; ++p) bounds[p] = (int)((long long)b->D * p / n_passes);
            }
            for (int p = 0; p < n_passes; ++p)
                if (bounds[p + 1] > bounds[p]) specs.push_back(PassSpec{bounds[p], bounds[p + 1], b->D, p, bounds[p + 1] >= b->D, nullptr, nullptr});
        } else {
            // small and medium batches: few launches matter more than the last few % of skipped samples
            // (tools/gpu_auto_probe.py: 1 object 32.3 ms vs 32.9 fixed-2; 8 objects 45.4 obj/s vs 43.1 fixed-10)
            const int n_passes = tiles >= 12.0 * h->n_cu ? 3 : 2;
            for (int p = 0; p < n_passes; ++p)
                specs.push_back(PassSpec{b->hint_margin, b->hint_step, b->D, p, p == n_passes - 1, b->ray_hint.p, b->ray_plo.p});
        }
        const int fwd_tile = pm ? LP_TILE_PTS : (use_split_fwd(b) ? SPLIT_TILE_PTS : TILE_PTS);
        const bool whole = pm && specs.size() == 1 && !specs[0].hint && specs[0].j0 == 0 && specs[0].j1 >= b->D;
        if (whole) {     // every in-sphere sample, in place: no selection list
            launch_build_tiles(b->oc.p, b->st.p, B, 0, b->tiles_f.p, b->n_tiles.p, b->counters.p, 1, fwd_tile, 4, s);
            launch_decoder(b, 3, cursor, true);
            specs.clear();
        }
        for (const PassSpec& ps : specs) {
            launch_pass_select(b->oc.p, b->st.p, b->raymask.p, b->rayoff.p, b->ray_alive.p, b->pcnt.p, ps, b->maxR, B, s);
            launch_scan_rays(b->oc.p, b->st.p, b->pcnt.p, b->poff.p, 2, B, s);
            launch_pass_write(b->oc.p, b->st.p, b->raymask.p, b->rayoff.p, b->ray_alive.p, b->poff.p, b->plist.p, ps, b->maxR, B, s);
            launch_build_tiles(b->oc.p, b->st.p, B, 2, b->tiles_f.p, b->n_tiles.p, b->counters.p, ps.pass == 0 ? 1 : 0, fwd_tile, pm ? 4 : 0, s);
            launch_decoder(b, pm ? 3 : 0, cursor);
            if (!ps.last || ps.hint)
                launch_pass_update(b->oc.p, b->st.p, b->raymask.p, b->rayoff.p, b->ray_alive.p, b->ssdf.p, pm ? thd : b->prm.cut_off, ps, b->maxR, B, s);
        }
        if (pm) {
            if (b->prepass_audit) {
                b->saudit.ensure(b->cap_s);
                launch_build_tiles(b->oc.p, b->st.p, B, 0, b->tiles_j.p, b->n_tiles.p + 3, b->counters.p, 0, TILE_PTS, 5, s);
                launch_decoder(b, 4, cursor);
                launch_prepass_audit(b->oc.p, b->st.p, b->ssdf.p, b->saudit.p, b->prm.cut_off, thd, b->audit_out.p, B, s);
            }
            if (spec) {
                b->srow.ensure(b->cap_s);
                b->jrow.ensure(b->cap_j);
                launch_band_fused(b->oc.p, b->st.p, b->raymask.p, b->rayoff.p, b->ssdf.p, thd, b->pcnt.p, b->poff.p, b->plist.p, b->spts.p, b->jpts.p,
                                  b->srow.p, B, s);
                // forward + backward of surface points and band samples in one launch; the band samples' sdf lands in ssdf
                launch_build_tiles(b->oc.p, b->st.p, B, 3, b->tiles_j.p, b->n_tiles.p + 1, b->counters.p, 0, use_split_rows(b) ? SPLIT_TILE_PTS : TILE_PTS, 1, s);
                launch_decoder(b, 1, cursor);
            } else {
                if (fused) launch_band_fused(b->oc.p, b->st.p, b->raymask.p, b->rayoff.p, b->ssdf.p, thd, b->pcnt.p, b->poff.p, b->plist.p, nullptr, nullptr, nullptr, B, s);
                else launch_band_select(b->oc.p, b->st.p, b->raymask.p, b->rayoff.p, b->ssdf.p, thd, b->pcnt.p, b->poff.p, b->plist.p, b->maxR, B, s);
                launch_build_tiles(b->oc.p, b->st.p, B, 2, b->tiles_f.p, b->n_tiles.p, b->counters.p, 0, use_split_fwd(b) ? SPLIT_TILE_PTS : TILE_PTS, 0, s);
                launch_decoder(b, 0, cursor);
            }
        }
        launch_render_scan(b->oc.p, b->st.p, b->raymask.p, b->rayoff.p, b->ssdf.p, b->depth.p, b->sdeds.p, b->ray_res.p,
                           b->kcnt.p, b->mcnt.p, b->D, b->prm.cut_off, b->maxR, B, s);
        if (fused) {
            launch_render_tail_fused(b->oc.p, b->st.p, b->raycnt.p, b->rayoff.p, b->spts.p, b->sdeds.p, b->ray_res.p, b->kcnt.p, b->koff.p, b->mcnt.p,
                                     b->jpts.p, b->jaux.p, spec ? b->srow.p : nullptr, spec ? b->jrow.p : nullptr, B, s);
        } else {
            launch_scan_rays(b->oc.p, b->st.p, b->kcnt.p, b->koff.p, 1, B, s);
            launch_sum_m(b->oc.p, b->st.p, b->mcnt.p, B, s);
            launch_render_write(b->oc.p, b->st.p, b->raycnt.p, b->rayoff.p, b->koff.p, b->spts.p, b->sdeds.p, b->ray_res.p,
                                b->jpts.p, b->jaux.p, b->maxR, B, s);
        }
    }
    if (spec) return;       // the jacobian launch already ran, ahead of the occupancy scan
    if (!fused) launch_surface(b->oc.p, b->st.p, b->pts.p, b->jpts.p, b->jaux.p, b->maxM, B, s);
    launch_build_tiles(b->oc.p, b->st.p, B, 1, b->tiles_j.p, b->n_tiles.p + 1, b->counters.p, 0, use_split_rows(b) ? SPLIT_TILE_PTS : TILE_PTS, 1, s);
    launch_decoder(b, 1, cursor);
    if (do_render && use_mask_reuse(b)) launch_decoder(b, 2, cursor);
}

void batch_run(dsp_batch* b) {
    dsp_handle* h = b->h;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int B = b->B;
    const int iters = b->pose_only ? b->prm.pose_only_iterations : b->prm.num_iterations;
    if (b->trace_on) b->trace.ensure((size_t)std::max(iters, 1) * B * TRACE_STRIDE);
    size_t cursor = 0;
    b->ev_kind.clear();
    hipEvent_t e_start = next_event(b, cursor);
    HIP_TRY(hipEventRecord(e_start, s));
    HIP_TRY(hipMemsetAsync(b->counters.p, 0, 8 * sizeof(double), s));
    HIP_TRY(hipMemsetAsync(b->audit_out.p, 0, 4 * sizeof(unsigned), s));
    HIP_TRY(hipMemsetAsync(b->st.p, 0, (size_t)B * sizeof(ObjState), s));
    launch_init_state(b->st.p, b->t_in.p, b->have_codes ? b->codes_in.p : nullptr, b->scale_in.p, B, b->D, b->pose_only ? 1 : 0, s);
    if (b->pose_only) HIP_TRY(hipMemsetAsync(b->alive.p, 1, b->cap_j, s));
    else HIP_TRY(hipMemsetAsync(b->ray_hint.p, b->D / 2, b->sum_rays, s));   // no history yet: first guess = object centre
    const GnParamsDev dp = dev_params(b);
    for (int e = 0; e < iters; ++e) {
        iteration_front(b, cursor, !b->pose_only, e == 0);
        launch_gram(b->oc.p, b->st.p, b->jpts.p, b->jaux.p, b->jgrad.p, use_speculative_band(b) ? b->jrow.p : nullptr, b->pose_only ? b->alive.p : nullptr, b->partials.p,
                    b->n_slices, b->prm.b2, b->prm.b1, b->pose_only ? 0 : 1, b->pose_only ? 1 : 2, B, s);
        launch_solve(b->oc.p, b->st.p, b->partials.p, b->gsum.p, b->n_slices, dp, e, b->trace_on ? b->trace.p : nullptr, h->codew.p, h->b0.p, h->blat.p,
                     b->cbias.p, B, s);
        if (b->pose_only && e == 4) launch_inlier_filter(b->oc.p, b->st.p, b->jgrad.p, b->alive.p, b->maxM, B, s);
    }
    launch_finalize(b->st.p, b->scale_in.p, B, b->pose_only ? 1 : 0, b->out_t.p, b->out_code.p, b->out_loss.p, b->out_status.p, s);
    hipEvent_t e_end = next_event(b, cursor);
    HIP_TRY(hipEventRecord(e_end, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    b->iters_run = iters;
    // stats
    dsp_stats st;
    memset(&st, 0, sizeof st);
    double cnt[8];
    HIP_TRY(hipMemcpy(cnt, b->counters.p, sizeof cnt, hipMemcpyDeviceToHost));
    unsigned aud[4];
    HIP_TRY(hipMemcpy(aud, b->audit_out.p, sizeof aud, hipMemcpyDeviceToHost));
    st.n_prepass_points = cnt[4];
    st.prepass_mode = prepass_mode(b);
    st.prepass_delta = st.prepass_mode ? prepass_delta(b) : 0.f;
    memcpy(&st.prepass_max_err, &aud[0], 4);
    st.prepass_misclassified = aud[1];
    st.prepass_audited = aud[2];
    st.n_fwd_points = cnt[0];
    const bool reuse = use_mask_reuse(b);
    st.n_jac_points = reuse ? cnt[1] : cnt[1] + cnt[3];
    st.n_insphere_points = cnt[2];
    st.n_render_rows = reuse ? cnt[3] : 0.0;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e_start, e_end));
    st.ms_total = ms;
    for (size_t i = 0; i < b->ev_kind.size(); ++i) {
        HIP_TRY(hipEventElapsedTime(&ms, b->ev[1 + 2 * i], b->ev[2 + 2 * i]));
        if (b->ev_kind[i] == 1) { st.ms_mlp_jac += ms; st.n_mlp_jac_launches++; }
        else if (b->ev_kind[i] == 2) { st.ms_mlp_prepass += ms; st.n_mlp_prepass_launches++; }
        else { st.ms_mlp_fwd += ms; st.n_mlp_fwd_launches++; }
    }
    b->stats = st;
}

void batch_results(dsp_batch* b, float* t_out, float* codes_out, float* loss_out, int32_t* status_out) {
    const int B = b->B;
    HIP_TRY(hipSetDevice(b->h->device));
    if (t_out) HIP_TRY(hipMemcpy(t_out, b->out_t.p, (size_t)B * 64, hipMemcpyDeviceToHost));
    if (codes_out) HIP_TRY(hipMemcpy(codes_out, b->out_code.p, (size_t)B * CODE_LEN * 4, hipMemcpyDeviceToHost));
    if (loss_out) HIP_TRY(hipMemcpy(loss_out, b->out_loss.p, (size_t)B * 4, hipMemcpyDeviceToHost));
    if (status_out) HIP_TRY(hipMemcpy(status_out, b->out_status.p, (size_t)B * 4, hipMemcpyDeviceToHost));
}

template <class F>
int guarded(dsp_handle* h, F f) {
    // one call at a time per handle: its HIP stream, scratch buffers and last mesh are shared state, and ctypes / pybind11
    // callers release the GIL around these calls, so two Python threads can arrive here together
    std::unique_lock<std::mutex> lock;
    if (h) lock = std::unique_lock<std::mutex>(h->mu);
    try {
        f();
        return DSP_OK;
    } catch (const std::invalid_argument& e) {
        if (h) h->err = e.what(); else g_create_error = e.what();
        return DSP_E_ARG;
    } catch (const std::bad_alloc& e) {
        if (h) h->err = "out of host memory"; else g_create_error = "out of host memory";
        return DSP_E_NOMEM;
    } catch (const std::exception& e) {
        if (h) h->err = e.what(); else g_create_error = e.what();
        return DSP_E_HIP;
    }
}

// stand-alone residual terms: a one-object batch whose state is set from the caller's t_obj_cam / depths
void run_terms(dsp_handle* h, const float* pts_cam, int64_t n_pts, const float* rays, int64_t n_rays, const float* depth_obs,
               const float* t_obj_cam, const float* sampled, int n_depths, const float* code, float th, int term,
               int64_t* k_out, float* jac_pose, float* jac_code, float* res, int64_t* v_out, int64_t* m_out) {
    dsp_gn_params prm;
    memset(&prm, 0, sizeof prm);
    prm.num_iterations = 1; prm.num_depth_samples = std::max(n_depths, 2); prm.cut_off = th; prm.lr = 1.f;
    const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const float dummy_pt[3] = {0, 0, 0};
    int64_t po[2] = {0, n_pts}, ro[2] = {0, n_rays}, dof[2] = {0, n_rays};
    const bool render = term == 1;
    if (render) { po[1] = 1; pts_cam = dummy_pt; }
    std::unique_ptr<dsp_batch> b(batch_build(h, &prm, 1, po, pts_cam, ro, rays, dof, depth_obs, eye, nullptr, nullptr, !render));
    ObjState st;
    memset(&st, 0, sizeof st);
    memcpy(st.t_oc, t_obj_cam, 64);
    memcpy(st.code, code, CODE_LEN * 4);
    if (render) {
        memcpy(st.depths, sampled, n_depths * 4);
        st.dmin = sampled[0]; st.dmax = sampled[n_depths - 1];
    }
    st.n_alive = -1;
    HIP_TRY(hipMemcpyAsync(b->st.p, &st, sizeof st, hipMemcpyHostToDevice, h->stream));
    if (render) HIP_TRY(hipMemsetAsync(b->ray_hint.p, b->D, b->sum_rays, h->stream));   // no history: decode whole rays in pass 0
    HIP_TRY(hipMemsetAsync(b->counters.p, 0, 8 * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(b->audit_out.p, 0, 4 * sizeof(unsigned), h->stream));
    size_t cursor = 0;
    b->ev_kind.clear();
    (void)next_event(b.get(), cursor);
    iteration_front(b.get(), cursor, render);
    const int cap = render ? (int)(n_rays * n_depths) : (int)n_pts;
    b->rows.alloc((size_t)std::max(cap, 1) * 72);
    launch_jrows(b->oc.p, b->st.p, b->jpts.p, b->jaux.p, b->jgrad.p, (render && use_speculative_band(b.get())) ? b->jrow.p : nullptr, term, b->rows.p, std::max(cap, 1), h->stream);
    HIP_TRY(hipMemcpyAsync(&st, b->st.p, sizeof st, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipGetLastError());
    int64_t n = render ? st.K : n_pts;
    if (render) {
        if (v_out) *v_out = st.V;
        if (m_out) *m_out = st.m;
        if (st.status == DSP_STATUS_FEW) { *k_out = -1; return; }
        *k_out = n;
    }
    std::vector<float> rows((size_t)n * 72);
    if (n) HIP_TRY(hipMemcpy(rows.data(), b->rows.p, rows.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
        memcpy(jac_pose + i * 7, rows.data() + i * 72, 7 * 4);
        memcpy(jac_code + i * CODE_LEN, rows.data() + i * 72 + 7, CODE_LEN * 4);
        res[i] = rows[i * 72 + 71];
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int dsp_abi_version(void) { return 2; }

/* Development aid: {shader-clock ticks, 100 MHz wall ticks} spent by workgroup 0 of the last dsp_decode_sdf /
 * dsp_sdf_jacobian launch -> effective shader clock under load. */
int dsp_debug_last_clocks(dsp_handle* h, uint64_t* out4) {
    if (!h || !out4 || !h->s_clk.p) return DSP_E_ARG;
    return guarded(h, [&] { HIP_TRY(hipSetDevice(h->device)); HIP_TRY(hipMemcpy(out4, h->s_clk.p, 32, hipMemcpyDeviceToHost)); });
}

/* Development aid: run forward+gradient on <= 64 points and dump every pass's output slab of the tile:
 * slabs_out[pass][wave][reg 0..127][lane 0..63], n_pass passes. */
int dsp_debug_slabs(dsp_handle* h, const float* code, const float* pts, int n, float* slabs_out, float* grad_out) {
    if (!h || n < 1 || n > 64) return DSP_E_ARG;
    return guarded(h, [&] {
        HIP_TRY(hipSetDevice(h->device));
        std::vector<float4> p4(n);
        for (int i = 0; i < n; ++i) p4[i] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], 0.f);
        int4 tile = make_int4(0, n, 0, 0);
        int nt = 1;
        DevBuf<float> dbg, out;
        const size_t nd = (size_t)h->n_pass_all * 4 * 128 * 64;
        dbg.alloc(nd);
        out.alloc(64 * GRAD_STRIDE);
        h->s_pts.ensure(64); h->s_code.ensure(CODE_LEN); h->s_tiles.ensure(1); h->s_ntiles.ensure(1);
        HIP_TRY(hipMemcpy(h->s_pts.p, p4.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->s_code.p, code, CODE_LEN * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->s_tiles.p, &tile, sizeof tile, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->s_ntiles.p, &nt, 4, hipMemcpyHostToDevice));
        MlpArgs a = make_mlp_args(h, 2);
        std::vector<float> cb(2 * WIDTH);
        code_bias_host(h->h_codew, h->h_b0, h->h_blat, code, cb.data());
        h->s_cbias.ensure(2 * WIDTH);
        HIP_TRY(hipMemcpy(h->s_cbias.p, cb.data(), cb.size() * 4, hipMemcpyHostToDevice));
        a.n_tiles = h->s_ntiles.p; a.tiles = h->s_tiles.p; a.pts = h->s_pts.p; a.codes = h->s_code.p; a.code_stride = CODE_LEN;
        a.code_bias = h->s_cbias.p; a.code_bias_stride = 2 * WIDTH;
        a.out_sdf = out.p; a.out_grad = out.p; a.dbg = dbg.p;
        HIP_TRY(launch_mlp(2, a, 1, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(slabs_out, dbg.p, nd * 4, hipMemcpyDeviceToHost));
        if (grad_out) HIP_TRY(hipMemcpy(grad_out, out.p, (size_t)n * GRAD_STRIDE * 4, hipMemcpyDeviceToHost));
    });
}

/* Host-only: the per-object code bias (layer 0 and latent_in layer) exactly as the library computes it for single-shot
 * calls; out = 1024 floats. */
int dsp_debug_code_bias(const dsp_decoder_desc* decoder, const float* code, float* out) {
    if (!decoder || !code || !out) return DSP_E_ARG;
    return guarded(nullptr, [&] {
        PackedNet pn;
        pack_decoder_host(&pn, decoder);
        code_bias_host(pn.codew, pn.b0, pn.blat, code, out);
    });
}

/* Host-only: pack a decoder exactly as dsp_create does and copy the result out (tests emulate the kernel's
 * data flow on it without a GPU).  pass_out receives n_pass x 8 int32 {nog,nchunks,bias_row,relu,mask_slot,kind,chunk_base,0};
 * meta_out = {n_fwd, n_pass, chunks_fwd, chunks_all, n_bias_rows, wlast_row, w0_row, lat_tile, code_len}.  Call with NULL buffers to query sizes. */
int dsp_debug_pack(const dsp_decoder_desc* decoder, float* stream_out, int64_t* stream_len, float* bias_out,
                   int64_t* bias_len, int32_t* pass_out, int32_t* meta_out, float* b_last_out) {
    if (!decoder || !stream_len || !bias_len || !meta_out) return DSP_E_ARG;
    return guarded(nullptr, [&] {
        PackedNet pn;
        pack_decoder_host(&pn, decoder);
        *stream_len = (int64_t)pn.stream.size();
        *bias_len = (int64_t)pn.bias.size();
        meta_out[0] = pn.n_fwd; meta_out[1] = pn.n_pass_all; meta_out[2] = pn.chunks_fwd; meta_out[3] = pn.chunks_all; meta_out[4] = pn.n_bias_rows;
        if (b_last_out) *b_last_out = pn.b_last;
        meta_out[5] = pn.wlast_row; meta_out[6] = pn.w0_row; meta_out[7] = pn.lat_tile; meta_out[8] = pn.code_len;
        if (stream_out) memcpy(stream_out, pn.stream.data(), pn.stream.size() * 4);
        if (bias_out) memcpy(bias_out, pn.bias.data(), pn.bias.size() * 4);
        if (pass_out)
            for (int i = 0; i < pn.n_pass_all; ++i) {
                const PassDesc& p = pn.pass[i];
                int32_t* o = pass_out + 8 * i;
                o[0] = p.nog; o[1] = p.nchunks; o[2] = p.bias_row; o[3] = p.relu; o[4] = p.mask_slot; o[5] = p.kind; o[6] = p.chunk_base; o[7] = 0;
            }
    });
}

int dsp_create(const dsp_decoder_desc* decoder, int device, dsp_handle** out) {
    if (!decoder || !out) { g_create_error = "null argument"; return DSP_E_ARG; }
    *out = nullptr;
    dsp_handle* h = nullptr;
    int rc = guarded(nullptr, [&] {
        int ndev = 0;
        HIP_TRY(hipGetDeviceCount(&ndev));
        if (device < 0 || device >= ndev) throw std::invalid_argument("no such HIP device");
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw std::runtime_error(std::string("libdspgn is built for gfx950 (MI355X); device is ") + prop.gcnArchName);
        h = new dsp_handle;
        h->device = device;
        h->n_cu = prop.multiProcessorCount;
        HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        HIP_TRY(mlp_prepare_device());
        HIP_TRY(mlp_split_prepare_device());
        HIP_TRY(mlp_lp_prepare_device());
        pack_decoder(h, decoder);
        calibrate_prepass(h);
    });
    if (rc != DSP_OK) { delete h; return rc; }
    *out = h;
    return DSP_OK;
}

void dsp_destroy(dsp_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
}

const char* dsp_last_error(const dsp_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dsp_decode_sdf(dsp_handle* h, const float* code, const float* pts, int64_t n, float* sdf_out) {
    if (!h || !code || (n > 0 && (!pts || !sdf_out)) || n < 0) return DSP_E_ARG;
    return guarded(h, [&] { run_decoder_points(h, code, 1, pts, n, false, sdf_out, nullptr); });
}

int dsp_decode_sdf_prepass(dsp_handle* h, int dtype, const float* code, const float* pts, int64_t n, float* sdf_out) {
    if (!h || !code || (n > 0 && (!pts || !sdf_out)) || n < 0 || (dtype != DSP_PREPASS_F16 && dtype != DSP_PREPASS_BF16)) return DSP_E_ARG;
    return guarded(h, [&] { run_decoder_points(h, code, 1, pts, n, false, sdf_out, nullptr, dtype); });
}

/* Host-only: pack the prepass weight stream exactly as dsp_create does (tests emulate the kernel's data flow on it without a GPU).
 * pass_out receives n_pass x 8 int32 {nog, nchunks, bias_row, kind, npad, last, chunk_base, 0}; meta_out = {n_pass, chunks}.
 * Call with NULL stream_out to query *stream_len (in 16-bit elements). */
int dsp_debug_pack_prepass(const dsp_decoder_desc* decoder, int dtype, uint16_t* stream_out, int64_t* stream_len, int32_t* pass_out,
                           int32_t* meta_out) {
    if (!decoder || !stream_len || !meta_out || (dtype != DSP_PREPASS_F16 && dtype != DSP_PREPASS_BF16)) return DSP_E_ARG;
    return guarded(nullptr, [&] {
        PackedLp pl;
        if (!pack_decoder_lp_host(&pl, decoder, dtype == DSP_PREPASS_BF16)) throw std::invalid_argument("decoder geometry not supported by the prepass kernel");
        *stream_len = (int64_t)pl.stream.size();
        meta_out[0] = pl.n_pass; meta_out[1] = pl.chunks;
        if (stream_out) memcpy(stream_out, pl.stream.data(), pl.stream.size() * 2);
        if (pass_out)
            for (int i = 0; i < pl.n_pass; ++i) {
                const LpPass& q = pl.pass[i];
                int32_t* o = pass_out + 8 * i;
                o[0] = q.nog; o[1] = q.nchunks; o[2] = q.bias_row; o[3] = q.kind; o[4] = q.npad; o[5] = q.last; o[6] = q.chunk_base; o[7] = 0;
            }
    });
}

int dsp_decode_sdf_multi(dsp_handle* h, const float* codes, int64_t n_codes, const float* pts, int64_t n, float* sdf_out) {
    if (!h || !codes || n_codes < 0 || n < 0 || (n > 0 && n_codes > 0 && (!pts || !sdf_out))) return DSP_E_ARG;
    return guarded(h, [&] { run_decoder_points(h, codes, n_codes, pts, n, false, sdf_out, nullptr); });
}

int dsp_extract_mesh(dsp_handle* h, const float* code, int32_t vol_dim, int32_t flags, int64_t* n_vertices, int64_t* n_faces) {
    if (!h || !code || vol_dim < 2 || vol_dim > 512 || !n_vertices || !n_faces || (flags & ~DSP_MESH_REGULAR_GRID)) return DSP_E_ARG;
    return guarded(h, [&] {
        HIP_TRY(hipSetDevice(h->device));
        const int64_t n = (int64_t)vol_dim * vol_dim * vol_dim;
        const float voxel_size = (float)(2.0 / (vol_dim - 1));
        h->s_pts.ensure(n);
        HIP_TRY(launch_grid_points(h->s_pts.p, vol_dim, voxel_size, (flags & DSP_MESH_REGULAR_GRID) ? 1 : 0, h->stream));
        decode_resident_points(h, code, 1, n, false);
        extract_mesh_device(h, h->s_out.p, vol_dim, vol_dim, vol_dim, 0.f, voxel_size, -1.f);
        *n_vertices = h->mesh_nv;
        *n_faces = h->mesh_nf;
    });
}

int dsp_marching_cubes(dsp_handle* h, const float* volume, int32_t n0, int32_t n1, int32_t n2, float level, float spacing, float origin,
                       int64_t* n_vertices, int64_t* n_faces) {
    if (!h || !volume || n0 < 2 || n1 < 2 || n2 < 2 || !n_vertices || !n_faces) return DSP_E_ARG;
    return guarded(h, [&] {
        HIP_TRY(hipSetDevice(h->device));
        const size_t n = (size_t)n0 * n1 * n2;
        h->mc_vol.ensure(n);
        HIP_TRY(hipMemcpyAsync(h->mc_vol.p, volume, n * 4, hipMemcpyHostToDevice, h->stream));
        extract_mesh_device(h, h->mc_vol.p, n0, n1, n2, level, spacing, origin);
        *n_vertices = h->mesh_nv;
        *n_faces = h->mesh_nf;
    });
}

int dsp_mesh_fetch(dsp_handle* h, float* vertices, int64_t n_vertices, int32_t* faces, int64_t n_faces) {
    if (!h || n_vertices < 0 || n_faces < 0 || (n_vertices > 0 && !vertices) || (n_faces > 0 && !faces)) return DSP_E_ARG;
    int state = 0;
    const int rc = guarded(h, [&] {
        // the counts are read under the handle's lock and must be the ones the caller sized its buffers for: another thread's
        // dsp_extract_mesh between this caller's extract and fetch is reported, never copied into the wrong-sized buffers
        if (h->mesh_nv < 0 || h->mesh_nv != n_vertices || h->mesh_nf != n_faces) { state = 1; return; }
        HIP_TRY(hipSetDevice(h->device));
        if (h->mesh_nv > 0) HIP_TRY(hipMemcpyAsync(vertices, h->mc_verts.p, (size_t)h->mesh_nv * 12, hipMemcpyDeviceToHost, h->stream));
        if (h->mesh_nf > 0) HIP_TRY(hipMemcpyAsync(faces, h->mc_faces.p, (size_t)h->mesh_nf * 12, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
    if (rc == DSP_OK && state) { h->err = "no mesh of that size on this handle (another extract ran in between?)"; return DSP_E_STATE; }
    return rc;
}

// host-only: the per-wave chunk order of the latency-form weight stream.  layout12 = off[4] | len[4] | len_fwd[4]; chunk_ids may be
// null (size query: *n_chunks is set), else it receives *n_chunks ids.
int dsp_debug_split_layout(const dsp_decoder_desc* decoder, int32_t* layout12, int32_t* chunk_ids, int64_t* n_chunks) {
    if (!decoder || !layout12 || !n_chunks) return DSP_E_ARG;
    return guarded(nullptr, [&] {
        PackedNet pn;
        pack_decoder_host(&pn, decoder);
        int off[4], len[4], lf[4];
        const std::vector<int> order = split_chunk_order(pn, off, len, lf);
        for (int w = 0; w < 4; ++w) { layout12[w] = off[w]; layout12[4 + w] = len[w]; layout12[8 + w] = lf[w]; }
        if (chunk_ids) {
            if (*n_chunks < (int64_t)order.size()) throw std::invalid_argument("chunk_ids too small");
            for (size_t i = 0; i < order.size(); ++i) chunk_ids[i] = order[i];
        }
        *n_chunks = (int64_t)order.size();
    });
}

int dsp_debug_solve_clocks(dsp_handle* h, uint64_t* out8) {
    if (!h || !out8) return DSP_E_ARG;
    return guarded(h, [&] { HIP_TRY(hipSetDevice(h->device)); HIP_TRY(debug_solve_clocks(reinterpret_cast<unsigned long long*>(out8))); });
}

int dsp_debug_mc_table(uint8_t* n_tri, uint8_t* tri) {
    if (!n_tri || !tri) return DSP_E_ARG;
    return guarded(nullptr, [&] {
        McTables t;
        mc_build_tables(t);
        memcpy(n_tri, t.n_tri, 256);
        for (int c = 0; c < 256; ++c) memcpy(tri + c * 3 * MC_MAX_TRI, t.tri[c], 3 * MC_MAX_TRI);
    });
}

int dsp_sdf_jacobian(dsp_handle* h, const float* code, const float* pts, int64_t n, float* sdf_out, float* grad_out) {
    if (!h || !code || (n > 0 && !pts) || n < 0) return DSP_E_ARG;
    return guarded(h, [&] { run_decoder_points(h, code, 1, pts, n, true, sdf_out, grad_out); });
}

int dsp_compute_sdf_loss(dsp_handle* h, const float* pts_cam, int64_t n, const float* t_obj_cam, const float* code,
                         float* jac_pose, float* jac_code, float* res) {
    if (!h || !pts_cam || n <= 0 || !t_obj_cam || !code || !jac_pose || !jac_code || !res) return DSP_E_ARG;
    return guarded(h, [&] {
        run_terms(h, pts_cam, n, nullptr, 0, nullptr, t_obj_cam, nullptr, 0, code, 0.01f, 0, nullptr, jac_pose, jac_code, res, nullptr, nullptr);
    });
}

int dsp_compute_render_loss(dsp_handle* h, const float* rays, int64_t n_rays, const float* depth_obs, const float* t_obj_cam,
                            const float* sampled_depth, int32_t n_depths, const float* code, float th, int64_t* k_out,
                            float* jac_pose, float* jac_code, float* res, int64_t* v_out, int64_t* m_out) {
    if (!h || !rays || n_rays <= 0 || !depth_obs || !t_obj_cam || !sampled_depth || n_depths < 2 || n_depths > MAX_DEPTH_SAMPLES ||
        !code || !k_out || !jac_pose || !jac_code || !res)
        return DSP_E_ARG;
    return guarded(h, [&] {
        run_terms(h, nullptr, 0, rays, n_rays, depth_obs, t_obj_cam, sampled_depth, n_depths, code, th, 1, k_out, jac_pose, jac_code, res, v_out, m_out);
    });
}

int dsp_batch_create(dsp_handle* h, const dsp_gn_params* prm, int32_t n_objects, const int64_t* pts_off, const float* pts,
                     const int64_t* ray_off, const float* rays, const int64_t* depth_off, const float* depth,
                     const float* t_cam_obj_in, const float* codes_in, dsp_batch** out) {
    if (!h || !prm || !pts_off || !pts || !ray_off || !rays || !depth_off || !t_cam_obj_in || !out) return DSP_E_ARG;
    return guarded(h, [&] { *out = batch_build(h, prm, n_objects, pts_off, pts, ray_off, rays, depth_off, depth, t_cam_obj_in, codes_in, nullptr, false); });
}

int dsp_batch_run(dsp_batch* b) {
    if (!b) return DSP_E_ARG;
    return guarded(b->h, [&] { batch_run(b); });
}

int dsp_batch_results(dsp_batch* b, float* t_cam_obj_out, float* codes_out, float* loss_out, int32_t* status_out) {
    if (!b) return DSP_E_ARG;
    return guarded(b->h, [&] { batch_results(b, t_cam_obj_out, codes_out, loss_out, status_out); });
}

int dsp_batch_stats(dsp_batch* b, dsp_stats* out) {
    if (!b || !out) return DSP_E_ARG;
    *out = b->stats;
    return DSP_OK;
}

int dsp_batch_set_ray_passes(dsp_batch* b, int n_passes) {
    if (!b || n_passes < 0 || n_passes > MAX_DEPTH_SAMPLES) return DSP_E_ARG;
    b->n_ray_passes = n_passes;
    b->pass_bounds.clear();
    return DSP_OK;
}

int dsp_batch_set_ray_pass_bounds(dsp_batch* b, const int32_t* bounds, int n_passes) {
    if (!b || !bounds || n_passes < 1 || n_passes > MAX_DEPTH_SAMPLES) return DSP_E_ARG;
    if (bounds[0] != 0 || bounds[n_passes] != b->D) return DSP_E_ARG;
    for (int p = 0; p < n_passes; ++p) if (bounds[p + 1] < bounds[p]) return DSP_E_ARG;
    b->n_ray_passes = n_passes;
    b->pass_bounds.assign(bounds, bounds + n_passes + 1);
    return DSP_OK;
}

int dsp_prepass_calibration(dsp_handle* h, int dtype, float* max_err, float* delta) {
    if (!h || (dtype != DSP_PREPASS_F16 && dtype != DSP_PREPASS_BF16)) return DSP_E_ARG;
    if (!h->lp_ok) return DSP_E_STATE;
    if (max_err) *max_err = h->lp_err[dtype == DSP_PREPASS_BF16 ? 1 : 0];
    if (delta) *delta = h->lp_delta[dtype == DSP_PREPASS_BF16 ? 1 : 0];
    return DSP_OK;
}

int dsp_batch_set_prepass(dsp_batch* b, int mode, float delta) {
    if (!b || mode < -1 || mode > DSP_PREPASS_BF16 || !(delta < 0.5f)) return DSP_E_ARG;
    if (mode > 0 && !b->h->lp_ok) return DSP_E_ARG;
    b->prepass = mode;
    b->prepass_delta = delta;
    return DSP_OK;
}

int dsp_batch_set_prepass_audit(dsp_batch* b, int on) {
    if (!b) return DSP_E_ARG;
    b->prepass_audit = on != 0;
    return DSP_OK;
}

int dsp_batch_set_speculative_band(dsp_batch* b, int mode) {
    if (!b || mode < -1 || mode > 1) return DSP_E_ARG;
    b->speculative = mode;
    return DSP_OK;
}

int dsp_batch_set_fused_bookkeeping(dsp_batch* b, int mode) {
    if (!b || mode < -1 || mode > 1) return DSP_E_ARG;
    b->fused_bookkeeping = mode;
    return DSP_OK;
}

int dsp_batch_set_mask_reuse(dsp_batch* b, int mode) {
    if (!b || mode < -1 || mode > 1) return DSP_E_ARG;
    b->mask_reuse = mode;
    return DSP_OK;
}

int dsp_batch_set_split_rows(dsp_batch* b, int mode) {
    if (!b || mode < -1 || mode > 1) return DSP_E_ARG;
    b->split_rows = mode;
    return DSP_OK;
}

int dsp_batch_enable_trace(dsp_batch* b, int on) {
    if (!b) return DSP_E_ARG;
    b->trace_on = on != 0;
    return DSP_OK;
}

int dsp_batch_trace(dsp_batch* b, int32_t iteration, float* H, float* bvec, float* dx, int64_t* V, int64_t* m, int64_t* K,
                    float* t_obj_cam, float* code, uint32_t* set_sums, float* depths) {
    if (!b || !b->trace_on || iteration < 0 || iteration >= b->iters_run) return DSP_E_STATE;
    return guarded(b->h, [&] {
        const int B = b->B;
        std::vector<float> tr((size_t)B * TRACE_STRIDE);
        HIP_TRY(hipSetDevice(b->h->device));
        HIP_TRY(hipMemcpy(tr.data(), b->trace.p + (size_t)iteration * B * TRACE_STRIDE, tr.size() * 4, hipMemcpyDeviceToHost));
        const int n = b->pose_only ? 6 : 71;
        for (int i = 0; i < B; ++i) {
            const float* t = tr.data() + (size_t)i * TRACE_STRIDE;
            if (H) for (int r = 0; r < n; ++r) memcpy(H + ((size_t)i * n + r) * n, t + r * 71, n * 4);
            if (bvec) memcpy(bvec + (size_t)i * n, t + 71 * 71, n * 4);
            if (dx) memcpy(dx + (size_t)i * n, t + 71 * 71 + 71, n * 4);
            if (t_obj_cam) memcpy(t_obj_cam + (size_t)i * 16, t + 71 * 71 + 142, 64);
            if (code) memcpy(code + (size_t)i * CODE_LEN, t + 71 * 71 + 142 + 16, CODE_LEN * 4);
            if (V) V[i] = (int64_t)t[71 * 71 + 142 + 80];
            if (m) m[i] = (int64_t)t[71 * 71 + 142 + 81];
            if (K) K[i] = (int64_t)t[71 * 71 + 142 + 82];
            if (depths) memcpy(depths + (size_t)i * 64, t + 5280, 64 * 4);
            if (set_sums) {
                set_sums[2 * i + 0] = (uint32_t)t[71 * 71 + 142 + 83] | ((uint32_t)t[71 * 71 + 142 + 84] << 16);
                set_sums[2 * i + 1] = (uint32_t)t[71 * 71 + 142 + 85] | ((uint32_t)t[71 * 71 + 142 + 86] << 16);
            }
        }
    });
}

void dsp_batch_destroy(dsp_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->h->device);
    delete b;
}

int dsp_reconstruct_batch(dsp_handle* h, const dsp_gn_params* prm, int32_t n_objects, const int64_t* pts_off, const float* pts,
                          const int64_t* ray_off, const float* rays, const int64_t* depth_off, const float* depth,
                          const float* t_cam_obj_in, const float* codes_in, float* t_cam_obj_out, float* codes_out,
                          float* loss_out, int32_t* status_out) {
    if (!h || !prm || !pts_off || !pts || !ray_off || !rays || !depth_off || !t_cam_obj_in) return DSP_E_ARG;
    return guarded(h, [&] {
        std::unique_ptr<dsp_batch> b(batch_build(h, prm, n_objects, pts_off, pts, ray_off, rays, depth_off, depth, t_cam_obj_in, codes_in, nullptr, false));
        batch_run(b.get());
        batch_results(b.get(), t_cam_obj_out, codes_out, loss_out, status_out);
    });
}

int dsp_estimate_pose_batch(dsp_handle* h, const dsp_gn_params* prm, int32_t n_objects, const int64_t* pts_off, const float* pts,
                            const float* t_co_se3_in, const float* scale, const float* codes, float* t_co_se3_out) {
    if (!h || !prm || !pts_off || !pts || !t_co_se3_in || !scale || !codes || !t_co_se3_out) return DSP_E_ARG;
    return guarded(h, [&] {
        std::unique_ptr<dsp_batch> b(batch_build(h, prm, n_objects, pts_off, pts, nullptr, nullptr, nullptr, nullptr, t_co_se3_in, codes, scale, true));
        batch_run(b.get());
        batch_results(b.get(), t_co_se3_out, nullptr, nullptr, nullptr);
    });
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// multi-GPU gather of results over RCCL (single process, one handle per GPU) -- SURVEY 8(e)
// ------------------------------------------------------------------------------------------------
namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::vector<int> devs;               // communicators are cached for the last device list
    std::vector<ncclComm_t> comms;
    std::mutex mu;
};
Rccl g_rccl;

void rccl_load() {
    if (g_rccl.lib) return;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) throw std::runtime_error(std::string("cannot load librccl: ") + dlerror());
#define RCCL_SYM(field, sym)                                                                  \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, #sym));      \
    if (!g_rccl.field) throw std::runtime_error("librccl lacks " #sym)
    RCCL_SYM(CommInitAll, ncclCommInitAll);
    RCCL_SYM(CommDestroy, ncclCommDestroy);
    RCCL_SYM(GroupStart, ncclGroupStart);
    RCCL_SYM(GroupEnd, ncclGroupEnd);
    RCCL_SYM(Gather, ncclGather);
    RCCL_SYM(GetErrorString, ncclGetErrorString);
#undef RCCL_SYM
}

#define RCCL_TRY(expr)                                                                                           \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess) throw std::runtime_error(std::string(#expr " failed: ") + g_rccl.GetErrorString(r_)); \
    } while (0)

}  // namespace

extern "C" int dsp_gather_results(dsp_handle* const* handles, int32_t n_handles, const float* const* results, const int32_t* n_objects,
                                  float* out) {
    if (!handles || n_handles < 1 || !results || !n_objects || !out) return DSP_E_ARG;
    for (int i = 0; i < n_handles; ++i)
        if (!handles[i] || n_objects[i] < 0 || (n_objects[i] > 0 && !results[i])) return DSP_E_ARG;
    dsp_handle* root = handles[0];
    return guarded(root, [&] {
        std::lock_guard<std::mutex> lock(g_rccl.mu);
        rccl_load();
        std::vector<int> devs(n_handles);
        int n_max = 0;
        for (int i = 0; i < n_handles; ++i) {
            devs[i] = handles[i]->device;
            n_max = std::max(n_max, (int)n_objects[i]);
            for (int j = 0; j < i; ++j)
                if (devs[j] == devs[i]) throw std::invalid_argument("dsp_gather_results: one handle per GPU (two handles share a device)");
        }
        if (devs != g_rccl.devs) {
            for (ncclComm_t c : g_rccl.comms) (void)g_rccl.CommDestroy(c);
            g_rccl.comms.assign(n_handles, nullptr);
            g_rccl.devs.clear();
            RCCL_TRY(g_rccl.CommInitAll(g_rccl.comms.data(), n_handles, devs.data()));
            g_rccl.devs = devs;
        }
        const size_t block = (size_t)std::max(n_max, 1) * DSP_RESULT_WIDTH;     // uneven shards are padded to the largest
        std::vector<DevBuf<float>> send(n_handles);
        DevBuf<float> recv;
        for (int i = 0; i < n_handles; ++i) {
            HIP_TRY(hipSetDevice(devs[i]));
            send[i].alloc(block);
            HIP_TRY(hipMemsetAsync(send[i].p, 0, block * 4, handles[i]->stream));
            if (n_objects[i])
                HIP_TRY(hipMemcpyAsync(send[i].p, results[i], (size_t)n_objects[i] * DSP_RESULT_WIDTH * 4, hipMemcpyHostToDevice, handles[i]->stream));
            if (i == 0) recv.alloc(block * n_handles);
        }
        RCCL_TRY(g_rccl.GroupStart());       // the ONE collective of the path: every GPU's block to the first handle's GPU, over xGMI
        for (int i = 0; i < n_handles; ++i)
            RCCL_TRY(g_rccl.Gather(send[i].p, i == 0 ? recv.p : nullptr, block, ncclFloat, 0, g_rccl.comms[i], handles[i]->stream));
        RCCL_TRY(g_rccl.GroupEnd());
        for (int i = 0; i < n_handles; ++i) {
            HIP_TRY(hipSetDevice(devs[i]));
            HIP_TRY(hipStreamSynchronize(handles[i]->stream));
        }
        HIP_TRY(hipSetDevice(devs[0]));
        std::vector<float> host(block * n_handles);
        HIP_TRY(hipMemcpy(host.data(), recv.p, host.size() * 4, hipMemcpyDeviceToHost));
        size_t o = 0;
        for (int i = 0; i < n_handles; ++i) {
            memcpy(out + o, host.data() + (size_t)i * block, (size_t)n_objects[i] * DSP_RESULT_WIDTH * 4);
            o += (size_t)n_objects[i] * DSP_RESULT_WIDTH;
        }
    });
}

extern "C" void dsp_pack_results(int32_t n, const float* t_cam_obj, const float* codes, const float* loss, const int32_t* status, float* packed) {
    for (int32_t i = 0; i < n; ++i) {
        float* row = packed + (size_t)i * DSP_RESULT_WIDTH;
        memcpy(row, t_cam_obj + (size_t)i * 16, 64);
        memcpy(row + 16, codes + (size_t)i * CODE_LEN, CODE_LEN * 4);
        row[80] = loss[i];
        row[81] = (float)status[i];
    }
}
