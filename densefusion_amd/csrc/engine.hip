// Host-side engine of libdfusion_hip.so: parameter store in kernel-friendly layouts, workspace
// planning, and the launch sequences of PoseNet.forward / PoseRefineNet.forward / the fused
// "estimate poses" pipeline (PoseNet -> per-pixel selection -> refine loop), all on one stream with
// no host synchronisation, no allocation and no host<->device copies inside a forward call.
//
// Reference behaviour mirrored: lib/network.py:95-132 (PoseNet.forward), :187-206
// (PoseRefineNet.forward), lib/pspnet.py:64-77, lib/extractors.py:114-124, tools/eval_ycb.py:192-229.
// Batch extension: the reference evaluates one object per call (b = 0 hard-coded, network.py:123);
// every entry point here takes B same-sized objects and evaluates them independently.
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "igemm.h"
#include "layers.h"
#include "wino.h"
#include "pose.h"

namespace df {

// ------------------------------------------------------------------------------------------------
// parameter store
// ------------------------------------------------------------------------------------------------
struct ParamInfo {
  std::string key;
  int64_t shape[4];
  int ndim;
  int64_t numel() const { int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; }
};

struct Net {
  int kind = 0;   // 0 = PoseNet, 1 = PoseRefineNet
  int num_points = 0, num_obj = 0;
  std::vector<ParamInfo> spec;
  std::map<std::string, int> index;
  std::vector<char> loaded;
  std::map<std::string, float *> buf;   // packed device buffers by internal name
  int device = 0;
  // profiling of GEMM launches (bench.py roofline): event pairs around every launch_conv
  bool profiling = false;
  std::vector<hipEvent_t> ev;
  std::vector<double> ev_flops, ev_bytes, ev_useful;
  std::vector<std::string> ev_desc;
  size_t ev_used = 0;
  // debug taps (df_net_debug_taps): copies of named intermediates of the last single-bucket forward, channels-last
  bool psp_dirty = false;               // a psp.* weight changed since the folded matrices were built
  // parameter uploads are enqueued on the null stream WITHOUT host synchronisation (one staging buffer, no per-tensor malloc /
  // free / sync); the next forward call waits for them once (check_ready)
  bool upload_pending = false;
  float *stage = nullptr;
  size_t stage_cap = 0;
  bool taps_on = false;
  struct Tap { float *buf = nullptr; size_t cap = 0; int64_t shape[4] = {0, 0, 0, 0}; };
  std::map<std::string, Tap> taps;
};

static void add(Net &n, const std::string &key, std::initializer_list<int64_t> shp) {
  ParamInfo p;
  p.key = key;
  p.ndim = (int)shp.size();
  int i = 0;
  for (auto v : shp) p.shape[i++] = v;
  for (; i < 4; ++i) p.shape[i] = 1;
  n.index[key] = (int)n.spec.size();
  n.spec.push_back(p);
}

static const char *CNN = "cnn.model.module.";

static void build_posenet_spec(Net &n) {
  const std::string c = CNN;
  add(n, c + "feats.conv1.weight", {64, 3, 7, 7});
  int inpl = 64;
  const int planes_of[4] = {64, 128, 256, 512};
  for (int li = 1; li <= 4; ++li) {
    const int planes = planes_of[li - 1];
    for (int blk = 0; blk < 2; ++blk) {
      const int cin = blk == 0 ? inpl : planes;
      const std::string base = c + "feats.layer" + std::to_string(li) + "." + std::to_string(blk) + ".";
      add(n, base + "conv1.weight", {planes, cin, 3, 3});
      add(n, base + "conv2.weight", {planes, planes, 3, 3});
      if (blk == 0 && cin != planes) add(n, base + "downsample.0.weight", {planes, cin, 1, 1});
    }
    inpl = planes;
  }
  for (int s = 0; s < 4; ++s) add(n, c + "psp.stages." + std::to_string(s) + ".1.weight", {512, 512, 1, 1});
  add(n, c + "psp.bottleneck.weight", {1024, 2560, 1, 1});
  add(n, c + "psp.bottleneck.bias", {1024});
  const char *ups[3] = {"up_1", "up_2", "up_3"};
  const int up_in[3] = {1024, 256, 64}, up_out[3] = {256, 64, 64};
  for (int u = 0; u < 3; ++u) {
    add(n, c + ups[u] + ".conv.1.weight", {up_out[u], up_in[u], 3, 3});
    add(n, c + ups[u] + ".conv.1.bias", {up_out[u]});
    add(n, c + ups[u] + ".conv.2.weight", {1});
  }
  add(n, c + "final.0.weight", {32, 64, 1, 1});
  add(n, c + "final.0.bias", {32});
  add(n, c + "classifier.0.weight", {256, 256});   // dead weights (lib/pspnet.py:58-62): accepted, unused
  add(n, c + "classifier.0.bias", {256});
  add(n, c + "classifier.2.weight", {21, 256});
  add(n, c + "classifier.2.bias", {21});
  const char *fn[6] = {"conv1", "conv2", "e_conv1", "e_conv2", "conv5", "conv6"};
  const int fi[6] = {3, 64, 32, 64, 256, 512}, fo[6] = {64, 128, 64, 128, 512, 1024};
  for (int i = 0; i < 6; ++i) {
    add(n, std::string("feat.") + fn[i] + ".weight", {fo[i], fi[i], 1});
    add(n, std::string("feat.") + fn[i] + ".bias", {fo[i]});
  }
  const int hin[3] = {1408, 640, 256}, hout[3] = {640, 256, 128};
  const char *hs[3] = {"r", "t", "c"};
  for (int l = 0; l < 3; ++l)
    for (int h = 0; h < 3; ++h) {
      const std::string nm = "conv" + std::to_string(l + 1) + "_" + hs[h];
      add(n, nm + ".weight", {hout[l], hin[l], 1});
      add(n, nm + ".bias", {hout[l]});
    }
  const int per[3] = {4, 3, 1};
  for (int h = 0; h < 3; ++h) {
    const std::string nm = std::string("conv4_") + hs[h];
    add(n, nm + ".weight", {(int64_t)n.num_obj * per[h], 128, 1});
    add(n, nm + ".bias", {(int64_t)n.num_obj * per[h]});
  }
}

static void build_refiner_spec(Net &n) {
  const char *fn[6] = {"conv1", "conv2", "e_conv1", "e_conv2", "conv5", "conv6"};
  const int fi[6] = {3, 64, 32, 64, 384, 512}, fo[6] = {64, 128, 64, 128, 512, 1024};
  for (int i = 0; i < 6; ++i) {
    add(n, std::string("feat.") + fn[i] + ".weight", {fo[i], fi[i], 1});
    add(n, std::string("feat.") + fn[i] + ".bias", {fo[i]});
  }
  const int li[2] = {1024, 512}, lo[2] = {512, 128};
  const char *hs[2] = {"r", "t"};
  for (int l = 0; l < 2; ++l)
    for (int h = 0; h < 2; ++h) {
      const std::string nm = "conv" + std::to_string(l + 1) + "_" + hs[h];
      add(n, nm + ".weight", {lo[l], li[l]});
      add(n, nm + ".bias", {lo[l]});
    }
  const int per[2] = {4, 3};
  for (int h = 0; h < 2; ++h) {
    const std::string nm = std::string("conv3_") + hs[h];
    add(n, nm + ".weight", {(int64_t)n.num_obj * per[h], 128});
    add(n, nm + ".bias", {(int64_t)n.num_obj * per[h]});
  }
}

static float *dev_alloc(Net &n, const std::string &name, size_t floats) {
  auto it = n.buf.find(name);
  if (it != n.buf.end()) return it->second;
  float *p = nullptr;
  if (hipMalloc(&p, floats * sizeof(float)) != hipSuccess) return nullptr;
  hipMemset(p, 0, floats * sizeof(float));
  n.buf[name] = p;
  return p;
}

// OIHW -> O (H W) Ipad
__global__ void pack_oihw_kernel(const float *__restrict__ src, float *__restrict__ dst, int O, int I, int HW, int Ipad) {
  const long total = (long)O * HW * Ipad;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Ipad);
    const long r = i / Ipad;
    const int t = (int)(r % HW);
    const long o = r / HW;
    dst[i] = c < I ? src[(o * I + c) * HW + t] : 0.f;
  }
}

// psp fold (load time): wc[s][m][i] = sum_o wb[m][512 s + o] * ws[s][o][i], accumulated in fp64 and
// rounded once, so the folded weights carry no more error than a single fp32 rounding
__global__ void psp_fold_kernel(const float *__restrict__ wb, const float *__restrict__ ws, float *__restrict__ wc, int s) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // 0..511
  const int m = blockIdx.y;
  if (i >= 512) return;
  double acc = 0.0;
  for (int o = 0; o < 512; ++o) acc += (double)wb[(size_t)m * 2560 + s * 512 + o] * (double)ws[(size_t)o * 512 + i];
  wc[((size_t)s * 1024 + m) * 512 + i] = (float)acc;
}

static bool ends_with(const std::string &s, const char *suf) {
  const size_t l = strlen(suf);
  return s.size() >= l && s.compare(s.size() - l, l, suf) == 0;
}

// copies `src` (device, reference layout) into the packed store
static int load_param(Net &n, const std::string &key, const float *src, int64_t numel) {
  auto it = n.index.find(key);
  if (it == n.index.end()) return set_error(DF_ERR_ARG, "load_param: unexpected key '%s'", key.c_str());
  const ParamInfo &pi = n.spec[it->second];
  if (numel != pi.numel())
    return set_error(DF_ERR_ARG, "load_param: size mismatch for %s: got %lld elements, expected %lld", key.c_str(),
                     (long long)numel, (long long)pi.numel());
  if (!src) return set_error(DF_ERR_ARG, "load_param: null pointer for %s", key.c_str());
  hipSetDevice(n.device);
  n.upload_pending = true;
  auto copy = [&](float *dst, const float *s, size_t cnt) { return hipMemcpyAsync(dst, s, cnt * sizeof(float), hipMemcpyDefault, 0); };
  auto copy2d = [&](float *dst, size_t dld, const float *s, size_t sld, size_t width, size_t rows) {
    return hipMemcpy2DAsync(dst, dld * sizeof(float), s, sld * sizeof(float), width * sizeof(float), rows, hipMemcpyDefault, 0);
  };
  hipError_t e = hipSuccess;
  if (pi.ndim == 4 && !ends_with(key, "classifier.0.weight")) {
    const int O = (int)pi.shape[0], I = (int)pi.shape[1], HW = (int)(pi.shape[2] * pi.shape[3]);
    const int Ipad = (I + 3) / 4 * 4;
    float *dst = dev_alloc(n, key, (size_t)O * HW * Ipad);
    if (!dst) return set_error(DF_ERR_LAUNCH, "load_param: hipMalloc failed");
    if (HW == 1 && Ipad == I) e = copy(dst, src, (size_t)numel);
    else {
      // src may be a host pointer: stage through the net's device staging buffer (grown on demand; stream order keeps a tensor's
      // pack kernel ahead of the next tensor's copy into the same buffer)
      if (n.stage_cap < (size_t)numel) {
        if (n.stage) { hipStreamSynchronize(0); hipFree(n.stage); }
        n.stage = nullptr; n.stage_cap = 0;
        if (hipMalloc(&n.stage, (size_t)numel * sizeof(float)) != hipSuccess) return set_error(DF_ERR_LAUNCH, "hipMalloc failed");
        n.stage_cap = (size_t)numel;
      }
      e = copy(n.stage, src, (size_t)numel);
      hipLaunchKernelGGL(pack_oihw_kernel, dim3(256), dim3(256), 0, 0, n.stage, dst, O, I, HW, Ipad);
    }
    if (e == hipSuccess && key.find("feats.layer") != std::string::npos && HW == 9 && I >= 128 && I % 4 == 0 && O % 4 == 0) {
      // stride-1 3x3 convs of layer2 .. layer4 may run in the Winograd domain (wino_route decides per map size): transformed copies
      // [16][O][I] for F(2x2,3x3) and [36][O][I] for F(4x4,3x3), fp64 math
      float *U = dev_alloc(n, key + ".wino", (size_t)16 * O * I), *U4 = dev_alloc(n, key + ".wino4", (size_t)36 * O * I);
      if (!U || !U4) return set_error(DF_ERR_LAUNCH, "load_param: hipMalloc failed");
      launch_wino_weight(dst, U, O, I, 0, 2);
      launch_wino_weight(dst, U4, O, I, 0, 4);
    }
    if (e == hipSuccess && (key.find(".up_1.") != std::string::npos || key.find(".up_2.") != std::string::npos) && HW == 9) {
      // up_1 / up_2 run as low-resolution 1x1 products per tap: keep a tap-major copy [9][O][I] (up_3 runs as a plain
      // [576]-deep GEMM on chosen-pixel patches and uses the packed layout as is)
      float *tm = dev_alloc(n, key + ".tm", (size_t)9 * O * I);
      if (!tm) return set_error(DF_ERR_LAUNCH, "load_param: hipMalloc failed");
      launch_tapmajor(dst, tm, O, I, 0);
    }
  } else if (n.kind == 0 && key.rfind("conv1_", 0) == 0) {
    // head layer 1 of tower h: split [640][1408] into the per-point part (first 384 input channels =
    // pointfeat_1|pointfeat_2) and the broadcast global-feature part (last 1024), towers stacked r,t,c
    const int h = key[6] == 'r' ? 0 : key[6] == 't' ? 1 : 2;
    if (ends_with(key, ".weight")) {
      float *wpt = dev_alloc(n, "head1.wpt", (size_t)1920 * 384), *wg = dev_alloc(n, "head1.wg", (size_t)1920 * 1024);
      if (!wpt || !wg) return set_error(DF_ERR_LAUNCH, "hipMalloc failed");
      e = copy2d(wpt + (size_t)h * 640 * 384, 384, src, 1408, 384, 640);
      if (e == hipSuccess) e = copy2d(wg + (size_t)h * 640 * 1024, 1024, src + 384, 1408, 1024, 640);
    } else {
      float *b = dev_alloc(n, "head1.bias", 1920);
      e = copy(b + h * 640, src, 640);
    }
  } else if (n.kind == 0 && (key.rfind("conv2_", 0) == 0 || key.rfind("conv3_", 0) == 0)) {
    const int l = key[4] - '0';
    const int h = key[6] == 'r' ? 0 : key[6] == 't' ? 1 : 2;
    const int co = l == 2 ? 256 : 128, ci = l == 2 ? 640 : 256;
    const std::string nm = std::string("head") + std::to_string(l);
    if (ends_with(key, ".weight")) e = copy(dev_alloc(n, nm + ".w", (size_t)3 * co * ci) + (size_t)h * co * ci, src, (size_t)co * ci);
    else e = copy(dev_alloc(n, nm + ".bias", 3 * co) + h * co, src, co);
  } else if (n.kind == 1 && (key.rfind("conv1_", 0) == 0 || key.rfind("conv2_", 0) == 0)) {
    const int l = key[4] - '0';
    const int h = key[6] == 'r' ? 0 : 1;
    const int co = l == 1 ? 512 : 128, ci = l == 1 ? 1024 : 512;
    const std::string nm = std::string("fc") + std::to_string(l);
    if (ends_with(key, ".weight")) e = copy(dev_alloc(n, nm + ".w", (size_t)2 * co * ci) + (size_t)h * co * ci, src, (size_t)co * ci);
    else e = copy(dev_alloc(n, nm + ".bias", 2 * co) + h * co, src, co);
  } else {
    float *dst = dev_alloc(n, key, (size_t)numel);
    if (!dst) return set_error(DF_ERR_LAUNCH, "load_param: hipMalloc failed");
    e = copy(dst, src, (size_t)numel);
  }
  if (e == hipSuccess && n.kind == 1 && key == "feat.conv5.weight") {
    // PoseRefineNetFeat.conv5 reads [x1 | e1 | x2 | e2] (lib/network.py:160-163); the colour half (e1, e2) does not change
    // between refine iterations, so its product is formed once per object: split the columns into an xyz part
    // [512][x1 64 | x2 128] and a colour part [512][e1 64 | e2 128] (the engine keeps the point features in that order)
    const float *w5 = n.buf[key];
    float *wx = dev_alloc(n, "feat.conv5.wx", (size_t)512 * 192), *we = dev_alloc(n, "feat.conv5.we", (size_t)512 * 192);
    if (!wx || !we) return set_error(DF_ERR_LAUNCH, "load_param: hipMalloc failed");
    e = copy2d(wx, 192, w5, 384, 64, 512);
    if (e == hipSuccess) e = copy2d(wx + 64, 192, w5 + 128, 384, 128, 512);
    if (e == hipSuccess) e = copy2d(we, 192, w5 + 64, 384, 64, 512);
    if (e == hipSuccess) e = copy2d(we + 64, 192, w5 + 256, 384, 128, 512);
  }
  if (e != hipSuccess) return set_error(DF_ERR_LAUNCH, "load_param(%s): %s", key.c_str(), hipGetErrorString(e));
  n.loaded[it->second] = 1;
  // the folded PSP matrices depend on five tensors: rebuilt once, by the next forward call (ensure_derived), not per key
  if (n.kind == 0 && key.find(".psp.") != std::string::npos && !ends_with(key, ".bias")) n.psp_dirty = true;
  return DF_OK;
}

// (re)build what is derived from several parameters: the PSP fold (bottleneck x stage weights, fp64 accumulation)
static int ensure_derived(Net &n) {
  if (!n.psp_dirty) {
    if (n.upload_pending) {               // the uploads of df_net_load_param ran on the null stream: one wait per batch of loads
      if (hipStreamSynchronize(0) != hipSuccess) return set_error(DF_ERR_LAUNCH, "parameter upload failed: %s", hipGetErrorString(hipGetLastError()));
      n.upload_pending = false;
    }
    return DF_OK;
  }
  const std::string P = CNN;
  float *wc = dev_alloc(n, "psp.fold.w", (size_t)4 * 1024 * 512), *wf = dev_alloc(n, "psp.fold.wfeat", (size_t)1024 * 512);
  if (!wc || !wf) return set_error(DF_ERR_LAUNCH, "psp fold: hipMalloc failed");
  const float *wb = n.buf[P + "psp.bottleneck.weight"];
  hipMemcpy2DAsync(wf, 512 * sizeof(float), wb + 2048, 2560 * sizeof(float), 512 * sizeof(float), 1024, hipMemcpyDeviceToDevice, 0);
  for (int st = 0; st < 4; ++st)
    hipLaunchKernelGGL(psp_fold_kernel, dim3(2, 1024, 1), dim3(256), 0, 0, wb, n.buf[P + "psp.stages." + std::to_string(st) + ".1.weight"], wc, st);
  if (hipStreamSynchronize(0) != hipSuccess || check_launch("psp fold") != DF_OK) return DF_ERR_LAUNCH;
  n.psp_dirty = false;
  n.upload_pending = false;
  return DF_OK;
}

static int check_ready(const Net &n0) {
  Net &n = const_cast<Net &>(n0);
  for (size_t i = 0; i < n.spec.size(); ++i)
    if (!n.loaded[i]) return set_error(DF_ERR_STATE, "parameter '%s' was never loaded", n.spec[i].key.c_str());
  return ensure_derived(n);
}

// ------------------------------------------------------------------------------------------------
// forward plumbing
// ------------------------------------------------------------------------------------------------
struct Ctx {
  Net *net;
  hipStream_t st;
  bool dry;          // dry run: only measure the workspace
  char *base;
  size_t off = 0, cap = 0, peak = 0;   // peak: high-water mark (scratch regions are released and reused, see wino_conv)
  int err = DF_OK;

  float *f(size_t floats) { return reinterpret_cast<float *>(bytes(floats * sizeof(float))); }
  void *bytes(size_t b) {
    b = (b + 255) & ~size_t(255);
    void *p = dry ? nullptr : base + off;
    off += b;
    if (off > peak) peak = off;
    if (!dry && off > cap && err == DF_OK) err = set_error(DF_ERR_WORKSPACE, "workspace too small: need > %zu bytes, have %zu", off, cap);
    return p;
  }
  bool live() const { return !dry && err == DF_OK; }
  const float *w(const std::string &name) {
    auto it = net->buf.find(name);
    if (it == net->buf.end()) {
      if (err == DF_OK) err = set_error(DF_ERR_STATE, "missing packed parameter '%s'", name.c_str());
      return nullptr;
    }
    return it->second;
  }
  // `useful`: the fraction of the launch's rows that are not padding (points beyond N in a 128-padded object block, Winograd
  // tiles beyond the map edge) -- only for the profile's useful-FLOP tally
  // debug tap: keep a copy of an intermediate (allocates: debug mode only, never under graph capture)
  void tap(const char *name, const float *src, int64_t d0, int64_t d1, int64_t d2, int64_t d3) {
    if (!live() || !net->taps_on || !src) return;
    Net::Tap &t = net->taps[name];
    const size_t n = (size_t)d0 * d1 * d2 * d3;
    if (t.cap < n) {
      if (t.buf) hipFree(t.buf);
      t.buf = nullptr; t.cap = 0;
      if (hipMalloc(&t.buf, n * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return; }
      t.cap = n;
    }
    t.shape[0] = d0; t.shape[1] = d1; t.shape[2] = d2; t.shape[3] = d3;
    hipMemcpyAsync(t.buf, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
  }
  double pt_useful = 1.0;      // N / Npad of the per-point launches
  void pconv(const ConvParams &p) { conv(p, pt_useful); }
  void conv(const ConvParams &p, double useful = 1.0) {
    if (!live()) return;
    Net &n = *net;
    if (n.profiling) {
      if (n.ev_used + 2 > n.ev.size()) {
        const size_t old = n.ev.size();
        n.ev.resize(old + 256);
        for (size_t i = old; i < n.ev.size(); ++i) hipEventCreate(&n.ev[i]);
      }
      hipEventRecord(n.ev[n.ev_used], st);
    }
#ifdef DF_DEV
    ConvParams pc = p;
    pc.wgt_const = true;           // the engine's packed parameters: split_gemm_invalidate() on every load / destroy
    const int rc = launch_conv(pc, st);
#else
    const int rc = launch_conv(p, st);
#endif
    if (n.profiling) {
      hipEventRecord(n.ev[n.ev_used + 1], st);
      n.ev_flops.push_back(conv_flops(p));
      n.ev_bytes.push_back(conv_bytes(p));
      n.ev_useful.push_back(conv_flops(p) * useful);
      char d[160];
      snprintf(d, sizeof(d), "M=%ld N=%d K=%d k%dx%d s%d d%d z%d", (long)p.B * p.OH * p.OW, p.Cout, p.KH * p.KW * p.Cin, p.KH,
               p.KW, p.stride, p.dil, p.zcount);
      n.ev_desc.push_back(d);
      n.ev_used += 2;
    }
    if (rc != DF_OK) err = rc;
  }
};

static ConvParams point_gemm(const float *in, int in_ld, int in_coff, int cin, const float *w, const float *bias,
                             float *out, int out_ld, int out_coff, int cout, int rows, int act);

// dev switch (A/B runs): DF_POINT_UNFUSED = the K = 3 / 32 / 64 per-point layers as separate launches (round 2) instead of pointfeat.hip
static bool point_fused() {
  static const bool on = df::dev_getenv("DF_POINT_UNFUSED") == nullptr;
  return on;
}

// dev switches (A/B runs): DF_NO_WINOGRAD = direct convolutions only, DF_WINOGRAD_TILE = 2 keeps F(2x2,3x3) where a Winograd route pays
static int wino_route_for(int H, int W, int dil, int ci, int co) {
  static const bool off = df::dev_getenv("DF_NO_WINOGRAD") != nullptr;
  static const int force = df::dev_getenv("DF_WINOGRAD_TILE") ? atoi(df::dev_getenv("DF_WINOGRAD_TILE")) : 0;
  if (off) return 0;
  const int r = wino_route(H, W, dil, ci, co);
  return (r && force == 2) ? 2 : r;
}

static ConvParams point_gemm(const float *in, int in_ld, int in_coff, int cin, const float *w, const float *bias,
                             float *out, int out_ld, int out_coff, int cout, int rows, int act) {
  ConvParams p;
  p.in = in; p.wgt = w; p.bias = bias; p.out = out;
  p.B = rows; p.H = p.W = p.OH = p.OW = 1;
  p.Cin = cin; p.in_ld = in_ld; p.in_coff = in_coff;
  p.Cout = cout; p.out_ld = out_ld; p.out_coff = out_coff;
  p.act = act;
  return p;
}

static ConvParams conv2d(const float *in, int B, int H, int W, int cin, int in_ld, const float *w, const float *bias,
                         float *out, int OH, int OW, int cout, int out_ld, int out_coff, int k, int stride, int pad,
                         int dil, int act) {
  ConvParams p;
  p.in = in; p.wgt = w; p.bias = bias; p.out = out;
  p.B = B; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
  p.Cin = cin; p.in_ld = in_ld; p.Cout = cout; p.out_ld = out_ld; p.out_coff = out_coff;
  p.KH = p.KW = k; p.stride = stride; p.pad = pad; p.dil = dil; p.act = act;
  return p;
}

static inline int conv_out(int in, int k, int stride, int pad, int dil) { return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1; }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct PoseNetOut {
  float *out_r, *out_t, *out_c, *emb;   // caller buffers
  float *emb_pm = nullptr;              // [B][Npad][32] inside the workspace
};

// One crop-size bucket of a call: B same-sized objects.  A call evaluates a list of buckets: every launch whose arithmetic does
// not depend on the crop geometry (1x1 convs, the Winograd-domain products, the low-resolution up-conv products and the whole
// per-point part) runs ONCE over the rows of all buckets; only the direct kxk convolutions and the memory-bound glue kernels
// (transforms, pooling, interpolation) are launched per bucket, writing into / reading from the shared row-concatenated buffers.
struct Grp { int B, H, W; const float *img; };

// per-bucket map sizes and row offsets of one resolution level of the concatenated activations
struct Level {
  std::vector<int> h, w;
  std::vector<long> off;     // first pixel row of bucket i
  long rows = 0;
  void push(int B, int hh, int ww) { h.push_back(hh); w.push_back(ww); off.push_back(rows); rows += (long)B * hh * ww; }
};

// PSPNet colour branch (lib/pspnet.py:64-77 on top of lib/extractors.py:114-124) up to up_2's output: returns the concatenated
// half-resolution 64-channel maps and their level
static float *cnn_forward(Ctx &c, const std::vector<Grp> &gs, Level &half_lv) {
  const std::string P = CNN;
  const int nb = (int)gs.size();
  Level l0, l1;          // stem output, max-pool output
  for (const Grp &g : gs) {
    const int H1 = conv_out(g.H, 7, 2, 3, 1), W1 = conv_out(g.W, 7, 2, 3, 1);
    l0.push(g.B, H1, W1);
    l1.push(g.B, conv_out(H1, 3, 2, 1, 1), conv_out(W1, 3, 2, 1, 1));
  }
  float *x = c.f((size_t)l1.rows * 64);
  {
    const size_t mark = c.off;     // img4 / stem are dead after the pooling: their region is reused by the layers below
    for (int i = 0; i < nb; ++i) {
      const Grp &g = gs[i];
      float *img4 = c.f((size_t)g.B * g.H * g.W * 4);
      if (c.live()) launch_nchw3_to_nhwc4(g.img, img4, g.B, g.H, g.W, c.st);
      float *stem = c.f((size_t)g.B * l0.h[i] * l0.w[i] * 64);
      c.conv(conv2d(img4, g.B, g.H, g.W, 4, 4, c.w(P + "feats.conv1.weight"), nullptr, stem, l0.h[i], l0.w[i], 64, 64, 0, 7, 2, 3, 1, ACT_RELU));
      if (c.live()) launch_maxpool3s2(stem, x + l1.off[i] * 64, g.B, l0.h[i], l0.w[i], 64, l1.h[i], l1.w[i], c.st);
      if (nb == 1) c.tap("stem", stem, g.B, l0.h[i], l0.w[i], 64);
    }
    c.off = mark;
    (void)mark;
  }

  // conv3x3 (+ residual) + ReLU over all buckets: buckets whose geometry makes the Winograd-domain product pay share ONE z-batched
  // GEMM over the concatenation of their tiles (per-bucket transforms around it); the others run the direct implicit GEMM
  auto conv3x3 = [&](const float *in, const Level &li, int ci, const std::string &key, float *out, const Level &lo, int co, int stride, int dil,
                     const float *res) {
    std::vector<int> ws[2];        // buckets on the F(2x2) / F(4x4) route
    for (int i = 0; i < nb; ++i) {
      const int route = stride == 1 ? wino_route_for(li.h[i], li.w[i], dil, ci, co) : 0;
      if (route) { ws[route == 4].push_back(i); continue; }
      ConvParams p = conv2d(in + li.off[i] * ci, gs[i].B, li.h[i], li.w[i], ci, ci, c.w(key), nullptr, out + lo.off[i] * co, lo.h[i], lo.w[i], co,
                            co, 0, 3, stride, dil, dil, ACT_RELU);
      if (res) { p.res = res + lo.off[i] * co; p.res_ld = co; }
      c.conv(p);
    }
    for (int r = 0; r < 2; ++r) {
      if (ws[r].empty()) continue;
      const int m = r ? 4 : 2, nz = (m + 2) * (m + 2);
      std::vector<long> t0;
      long T = 0;
      for (int i : ws[r]) { t0.push_back(T); T += wino_geom(gs[i].B, li.h[i], li.w[i], dil, m).T; }
      const size_t mark = c.off;       // V / M are scratch: consecutive layers reuse the same region
      float *V = c.f((size_t)nz * T * ci), *M = c.f((size_t)nz * T * co);
      // per-bucket tables of the transforms: F(4x4) runs all buckets of the route in one launch (a workgroup belongs to one bucket)
      std::vector<int> tB, tH, tW;
      std::vector<long> trow;
      for (int i : ws[r]) { tB.push_back(gs[i].B); tH.push_back(li.h[i]); tW.push_back(li.w[i]); trow.push_back(li.off[i]); }
      const int nw = (int)ws[r].size();
      static const bool per_bucket = df::dev_getenv("DF_WINO_PER_BUCKET") != nullptr;       // dev switch: one transform launch per bucket (A/B)
      if (m == 4 && !per_bucket) {
        if (c.live()) launch_wino4_input_multi(in, ci, V, nw, tB.data(), tH.data(), tW.data(), trow.data(), t0.data(), ci, dil, T, c.st);
      } else {
        for (size_t j = 0; j < ws[r].size(); ++j) {
          const int i = ws[r][j];
          if (c.live()) launch_wino_input(in + li.off[i] * ci, ci, 0, V, gs[i].B, li.h[i], li.w[i], ci, dil, c.st, T, t0[j], m);
        }
      }
      ConvParams p = point_gemm(V, ci, 0, ci, c.w(key + (r ? ".wino4" : ".wino")), nullptr, M, co, 0, co, (int)T, ACT_NONE);
      p.zcount = nz; p.z_in_coff = T * ci; p.z_wgt = (long)co * ci; p.z_out_coff = T * co;
      double px = 0;        // output pixels the tiles are for: a tile yields m x m of them
      for (int i : ws[r]) px += (double)gs[i].B * li.h[i] * li.w[i];
      c.conv(p, px / ((double)(m * m) * (double)T));
      if (m == 4 && !per_bucket) {           // (stride-1 layers: input and output levels have the same rows)
        if (c.live()) launch_wino4_output_multi(M, out, co, res, co, ACT_RELU, nw, tB.data(), tH.data(), tW.data(), trow.data(), t0.data(), co, dil, T, c.st);
      } else {
        for (size_t j = 0; j < ws[r].size(); ++j) {
          const int i = ws[r][j];
          if (c.live())
            launch_wino_output(M, out + lo.off[i] * co, co, 0, nullptr, res ? res + lo.off[i] * co : nullptr, co, 0, ACT_RELU, gs[i].B, li.h[i], li.w[i],
                               co, dil, c.st, T, t0[j], m);
        }
      }
      c.off = mark;
    }
  };

  Level lx = l1;
  int cin = 64;
  const int planes_of[4] = {64, 128, 256, 512}, stride_of[4] = {1, 2, 1, 1}, dil_of[4] = {1, 1, 2, 4};
  for (int li = 1; li <= 4; ++li) {
    const int planes = planes_of[li - 1], s = stride_of[li - 1], d = dil_of[li - 1];
    const std::string base = P + "feats.layer" + std::to_string(li) + ".";
    Level lo;
    for (int i = 0; i < nb; ++i) lo.push(gs[i].B, conv_out(lx.h[i], 3, s, 1, 1), conv_out(lx.w[i], 3, s, 1, 1));
    // block 0: built without dilation (lib/extractors.py:107); carries the stride and the 1x1 downsample
    float *t = c.f((size_t)lo.rows * planes);
    conv3x3(x, lx, cin, base + "0.conv1.weight", t, lo, planes, s, 1, nullptr);
    const float *res = x;
    if (cin != planes || s != 1) {
      float *ds = c.f((size_t)lo.rows * planes);
      if (s == 1) {     // a 1x1 stride-1 conv is a plain GEMM over pixel rows: one launch for all buckets
        c.conv(point_gemm(x, cin, 0, cin, c.w(base + "0.downsample.0.weight"), nullptr, ds, planes, 0, planes, (int)lo.rows, ACT_NONE));
      } else {
        for (int i = 0; i < nb; ++i)
          c.conv(conv2d(x + lx.off[i] * cin, gs[i].B, lx.h[i], lx.w[i], cin, cin, c.w(base + "0.downsample.0.weight"), nullptr, ds + lo.off[i] * planes,
                        lo.h[i], lo.w[i], planes, planes, 0, 1, s, 0, 1, ACT_NONE));
      }
      res = ds;
    }
    float *o0 = c.f((size_t)lo.rows * planes);
    conv3x3(t, lo, planes, base + "0.conv2.weight", o0, lo, planes, 1, 1, res);
    // block 1: dilated (lib/extractors.py:110)
    float *t1 = c.f((size_t)lo.rows * planes);
    conv3x3(o0, lo, planes, base + "1.conv1.weight", t1, lo, planes, 1, d, nullptr);
    float *o1 = c.f((size_t)lo.rows * planes);
    conv3x3(t1, lo, planes, base + "1.conv2.weight", o1, lo, planes, 1, d, o0);
    x = o1; lx = lo; cin = planes;
    if (nb == 1) c.tap(("layer" + std::to_string(li)).c_str(), x, gs[0].B, lx.h[0], lx.w[0], planes);
  }
  // PSP module (lib/pspnet.py:20-24) with the bottleneck folded through the pyramid:
  //   bottleneck(cat(up(W_s pool_s(f)), f)) = W_b[:,2048:] f + sum_s up((W_b[:,512s:512s+512] W_s) pool_s(f)) + b
  // (1x1 convs and bilinear resampling are linear and commute), so the 2560-channel concat is never built,
  // the big GEMM shrinks from K=2560 to K=512 and the four stage convs become one grouped launch on
  // 50 pooled rows per object with weights combined once at load time (psp_fold).
  float *prior = c.f((size_t)lx.rows * 1024);
  for (int i = 0; i < nb; ++i) {
    const int B = gs[i].B;
    float *pooled = c.f((size_t)4 * B * 36 * 512), *zst = c.f((size_t)4 * B * 36 * 1024);
    if (c.live()) launch_psp_pool(x + lx.off[i] * 512, 512, 0, pooled, B, lx.h[i], lx.w[i], 512, c.st);
    ConvParams p = point_gemm(pooled, 512, 0, 512, c.w("psp.fold.w"), nullptr, zst, 1024, 0, 1024, B * 36, ACT_NONE);
    p.zcount = 4; p.z_in_coff = (long)B * 36 * 512; p.z_wgt = 1024 * 512; p.z_out_coff = (long)B * 36 * 1024;
    c.conv(p);
    if (c.live()) launch_psp_prior_sum(zst, prior + lx.off[i] * 1024, B, lx.h[i], lx.w[i], 1024, c.st);
  }
  float *psp = c.f((size_t)lx.rows * 1024);
  {
    ConvParams p = point_gemm(x, 512, 0, 512, c.w("psp.fold.wfeat"), c.w(P + "psp.bottleneck.bias"), psp, 1024, 0, 1024, (int)lx.rows, ACT_RELU);
    p.res = prior; p.res_ld = 1024;
    c.conv(p);
  }
  if (nb == 1) c.tap("psp", psp, gs[0].B, lx.h[0], lx.w[0], 1024);
  // PSPUpsample stages up_1, up_2 (lib/pspnet.py:27-37,69-73; dropout = identity in eval), each as a low-resolution
  // GEMM with N = 9*Cout (one launch for all buckets) followed by the per-bucket 9-tap interpolation (layers.hip).  up_3 is NOT run
  // here: its output is read at the chosen pixels only, so the caller evaluates it there from the maps returned: [B][h][w][64].
  float *cur = psp;
  const char *ups[2] = {"up_1", "up_2"};
  const int up_in[2] = {1024, 256}, up_out[2] = {256, 64};
  for (int u = 0; u < 2; ++u) {
    const size_t mark = c.off;
    float *o = c.f((size_t)4 * lx.rows * up_out[u]);
    const size_t keep = c.off;
    float *y = c.f((size_t)lx.rows * 9 * up_out[u]);
    c.conv(point_gemm(cur, up_in[u], 0, up_in[u], c.w(P + ups[u] + ".conv.1.weight.tm"), nullptr, y, 9 * up_out[u], 0, 9 * up_out[u], (int)lx.rows,
                      ACT_NONE));
    Level lo;
    for (int i = 0; i < nb; ++i) lo.push(gs[i].B, 2 * lx.h[i], 2 * lx.w[i]);
    for (int i = 0; i < nb; ++i)
      if (c.live()) {
        const int rc = launch_upconv_gather(y + lx.off[i] * 9 * up_out[u], c.w(P + ups[u] + ".conv.1.bias"), c.w(P + ups[u] + ".conv.2.weight"),
                                            o + lo.off[i] * up_out[u], gs[i].B, lx.h[i], lx.w[i], up_out[u], c.st);
        if (rc != DF_OK) c.err = rc;
      }
    c.off = keep;        // the tap products are dead once interpolated
    (void)mark;
    lx = lo;
    cur = o;
    if (nb == 1) c.tap(ups[u], cur, gs[0].B, lx.h[0], lx.w[0], up_out[u]);
  }
  half_lv = lx;
  return cur;
}

// PoseNetFeat + heads (lib/network.py:53-68,107-131) on point-major rows padded to Npad per object
// `sel` non-null (eval loop): only the confidence tower runs over all points; the r / t towers are evaluated at the
// arg-max point alone (launch_head_select), which also writes the pose record -- out_r / out_t / out_c stay untouched.
struct SelectOut { double *pose_wo, *state; float *rt; };

static void posenet_points(Ctx &c, int B, int N, int Npad, const float *cloud, const float *emb_pm, const int64_t *obj,
                           float *out_r, float *out_t, float *out_c, const SelectOut *sel = nullptr) {
  Net &n = *c.net;
  const int rows = B * Npad;
  float *pf = c.f((size_t)rows * 384);          // [x1 64 | e1 64 | x2 128 | e2 128] = pointfeat_1 | pointfeat_2
  if (point_fused()) {        // conv1 -> conv2 and e_conv1 -> e_conv2 chained in one launch, the 64-wide intermediates in LDS
    PointFeatParams q;
    q.cloud = cloud; q.emb = emb_pm; q.pf = pf; q.ld = 384; q.cx1 = 0; q.ce1 = 64; q.cx2 = 128; q.ce2 = 256;
    q.w1 = c.w("feat.conv1.weight"); q.b1 = c.w("feat.conv1.bias"); q.w2 = c.w("feat.conv2.weight"); q.b2 = c.w("feat.conv2.bias");
    q.we1 = c.w("feat.e_conv1.weight"); q.be1 = c.w("feat.e_conv1.bias"); q.we2 = c.w("feat.e_conv2.weight"); q.be2 = c.w("feat.e_conv2.bias");
    q.B = B; q.N = N; q.Npad = Npad;
    if (c.live()) { const int rc = launch_pointfeat(q, c.st); if (rc != DF_OK) c.err = rc; }
  } else {
    if (c.live()) launch_cloud_conv1(cloud, nullptr, c.w("feat.conv1.weight"), c.w("feat.conv1.bias"), pf, 384, B, N, Npad, c.st);
    c.pconv(point_gemm(emb_pm, 32, 0, 32, c.w("feat.e_conv1.weight"), c.w("feat.e_conv1.bias"), pf, 384, 64, 64, rows, ACT_RELU));
    c.pconv(point_gemm(pf, 384, 0, 64, c.w("feat.conv2.weight"), c.w("feat.conv2.bias"), pf, 384, 128, 128, rows, ACT_RELU));
    c.pconv(point_gemm(pf, 384, 64, 64, c.w("feat.e_conv2.weight"), c.w("feat.e_conv2.bias"), pf, 384, 256, 128, rows, ACT_RELU));
  }
  float *x5 = c.f((size_t)rows * 512);
  c.pconv(point_gemm(pf, 384, 128, 256, c.w("feat.conv5.weight"), c.w("feat.conv5.bias"), x5, 512, 0, 512, rows, ACT_RELU));
  // conv6 + ReLU + AvgPool1d(N): the 1024-wide activation is consumed only by the mean, so it never
  // leaves the GEMM's registers -- per-wave column sums, then a tiny deterministic reduction
  ConvParams p6 = point_gemm(x5, 512, 0, 512, c.w("feat.conv6.weight"), c.w("feat.conv6.bias"), nullptr, 1024, 0, 1024, rows, ACT_RELU);
  p6.rows_per_group = Npad; p6.rows_valid = N;
  const int prow = conv_colsum_rows(p6);
  float *partial = c.f((size_t)prow * 1024);
  p6.colsum = partial;
  c.pconv(p6);
  float *apx = c.f((size_t)B * 1024);
  if (c.live()) launch_colsum_finish(partial, prow / B, apx, B, 1024, N, c.st);
  c.tap("ap_x", apx, B, 1024, 1, 1);
  // head layer 1: W[:, :384] . pointfeat + (W[:, 384:] . ap_x + b) -- the 1024 broadcast channels of the
  // 1408-wide input are identical for every point of an object, so they collapse into a per-object bias
  float *gbias = c.f((size_t)B * 1920);
  if (c.live()) launch_fc_rows(apx, 1024, 0, c.w("head1.wg"), c.w("head1.bias"), gbias, 1920, B, 1024, 1920, 1, 0, c.st);   // one row per object
  if (sel) {
    float *h1c = c.f((size_t)rows * 640), *h2c = c.f((size_t)rows * 256), *h3c = c.f((size_t)rows * 128), *conf = c.f((size_t)B * N);
    const float *w1 = c.w("head1.wpt"), *w2 = c.w("head2.w"), *b2 = c.w("head2.bias"), *w3 = c.w("head3.w"), *b3 = c.w("head3.bias");
    if (c.err != DF_OK) return;
    {
      ConvParams p = point_gemm(pf, 384, 0, 384, c.dry ? nullptr : w1 + (size_t)2 * 640 * 384, c.dry ? nullptr : gbias + 1280, h1c, 640, 0, 640, rows, ACT_RELU);
      p.rows_per_group = Npad; p.rows_valid = N; p.bias_group_ld = 1920;
      c.pconv(p);
    }
    c.pconv(point_gemm(h1c, 640, 0, 640, c.dry ? nullptr : w2 + (size_t)2 * 256 * 640, c.dry ? nullptr : b2 + 512, h2c, 256, 0, 256, rows, ACT_RELU));
    c.pconv(point_gemm(h2c, 256, 0, 256, c.dry ? nullptr : w3 + (size_t)2 * 128 * 256, c.dry ? nullptr : b3 + 256, h3c, 128, 0, 128, rows, ACT_RELU));
    if (c.live())
      launch_head_select(h3c, c.w("conv4_c.weight"), c.w("conv4_c.bias"), pf, gbias, w1, w2, b2, w3, b3, c.w("conv4_r.weight"),
                         c.w("conv4_r.bias"), c.w("conv4_t.weight"), c.w("conv4_t.bias"), obj, n.num_obj, cloud, B, N, Npad, conf, sel->pose_wo,
                         sel->state, sel->rt, nullptr, c.st);
    return;
  }
  float *h1 = c.f((size_t)rows * 1920);
  {
    ConvParams p = point_gemm(pf, 384, 0, 384, c.w("head1.wpt"), gbias, h1, 1920, 0, 1920, rows, ACT_RELU);
    p.rows_per_group = Npad; p.rows_valid = N; p.bias_group_ld = 1920;
    c.pconv(p);
  }
  float *h2 = c.f((size_t)rows * 768), *h3 = c.f((size_t)rows * 384);
  {
    ConvParams p = point_gemm(h1, 1920, 0, 640, c.w("head2.w"), c.w("head2.bias"), h2, 768, 0, 256, rows, ACT_RELU);
    p.zcount = 3; p.z_in_coff = 640; p.z_wgt = 256 * 640; p.z_bias = 256; p.z_out_coff = 256;
    c.pconv(p);
  }
  {
    ConvParams p = point_gemm(h2, 768, 0, 256, c.w("head3.w"), c.w("head3.bias"), h3, 384, 0, 128, rows, ACT_RELU);
    p.zcount = 3; p.z_in_coff = 256; p.z_wgt = 128 * 256; p.z_bias = 128; p.z_out_coff = 128;
    c.pconv(p);
  }
  if (c.live())
    launch_head_final(h3, c.w("conv4_r.weight"), c.w("conv4_r.bias"), c.w("conv4_t.weight"), c.w("conv4_t.bias"),
                      c.w("conv4_c.weight"), c.w("conv4_c.bias"), obj, n.num_obj, out_r, out_t, out_c, B, N, Npad, c.st);
}

static void posenet_forward(Ctx &c, const std::vector<Grp> &gs, const float *cloud, const int64_t *choose, const int64_t *obj, PoseNetOut &o,
                            const SelectOut *sel = nullptr) {
  const int N = c.net->num_points, Npad = round_up(N, 128);
  c.pt_useful = (double)N / Npad;
  int B = 0;
  for (const Grp &g : gs) B += g.B;
  Level hl;
  float *half = cnn_forward(c, gs, hl);     // up_2's outputs: half resolution, 64 channels, buckets concatenated
  // up_3 (bilinear x2 + conv3x3 + PReLU) at the chosen pixels: patch rows (per bucket: the only geometry-dependent step), then one
  // GEMM over the points of all objects with the PReLU fused, then the final 1x1 conv + LogSoftmax
  const std::string P = CNN;
  float *patch = c.f((size_t)B * Npad * 576), *z3 = c.f((size_t)B * Npad * 64);
  {
    long b0 = 0;
    for (size_t i = 0; i < gs.size(); ++i) {
      if (c.live())
        launch_up3_patches(half + hl.off[i] * 64, choose + b0 * N, patch + (size_t)b0 * Npad * 576, gs[i].B, hl.h[i], hl.w[i], N, Npad, c.st);
      b0 += gs[i].B;
    }
  }
  {
    ConvParams p = point_gemm(patch, 576, 0, 576, c.w(P + "up_3.conv.1.weight"), c.w(P + "up_3.conv.1.bias"), z3, 64, 0, 64, B * Npad, ACT_PRELU);
    p.prelu = c.w(P + "up_3.conv.2.weight");
    c.pconv(p);
  }
  if (gs.size() == 1) c.tap("up_3", z3, B, Npad, 64, 1);      // rows of the chosen pixels only
  o.emb_pm = c.f((size_t)B * Npad * 32);
  if (c.live()) launch_final_logsoftmax(z3, c.w(P + "final.0.weight"), c.w(P + "final.0.bias"), o.emb, o.emb_pm, B, N, Npad, c.st);
  posenet_points(c, B, N, Npad, cloud, o.emb_pm, obj, o.out_r, o.out_t, o.out_c, sel);
}

// PoseRefineNetFeat + FC towers (lib/network.py:151-168,187-204).  The emb branch (e_conv1/e_conv2) does
// not depend on the cloud, so `prepare` runs it once per object and `iterate` re-does only the xyz branch.
struct RefinerBufs { float *pf, *e5, *x5, *partial, *apx, *f1, *f2; int prow; };   // pf rows: [x1 64 | x2 128 | e1 64 | e2 128]

static RefinerBufs refiner_alloc(Ctx &c, int B, int N, int Npad) {
  RefinerBufs r;
  c.pt_useful = (double)N / Npad;
  const int rows = B * Npad;
  r.pf = c.f((size_t)rows * 384);
  r.e5 = c.f((size_t)rows * 512);
  r.x5 = c.f((size_t)rows * 512);
  ConvParams p6 = point_gemm(r.x5, 512, 0, 512, nullptr, nullptr, nullptr, 1024, 0, 1024, rows, ACT_RELU);
  p6.rows_per_group = Npad; p6.rows_valid = N;
  r.prow = conv_colsum_rows(p6);
  r.partial = c.f((size_t)r.prow * 1024);
  r.apx = c.f((size_t)B * 1024);
  r.f1 = c.f((size_t)B * 1024);
  r.f2 = c.f((size_t)B * 256);
  return r;
}

static void refiner_prepare(Ctx &c, const RefinerBufs &r, int B, int Npad, const float *emb_pm) {
  const int rows = B * Npad;
  if (point_fused()) {
    PointFeatParams q;
    q.emb = emb_pm; q.pf = r.pf; q.ld = 384; q.ce1 = 192; q.ce2 = 256;
    q.we1 = c.w("feat.e_conv1.weight"); q.be1 = c.w("feat.e_conv1.bias"); q.we2 = c.w("feat.e_conv2.weight"); q.be2 = c.w("feat.e_conv2.bias");
    q.B = B; q.N = c.net->num_points; q.Npad = Npad;
    if (c.live()) { const int rc = launch_pointfeat(q, c.st); if (rc != DF_OK) c.err = rc; }
  } else {
    c.pconv(point_gemm(emb_pm, 32, 0, 32, c.w("feat.e_conv1.weight"), c.w("feat.e_conv1.bias"), r.pf, 384, 192, 64, rows, ACT_RELU));
    c.pconv(point_gemm(r.pf, 384, 192, 64, c.w("feat.e_conv2.weight"), c.w("feat.e_conv2.bias"), r.pf, 384, 256, 128, rows, ACT_RELU));
  }
  // colour half of conv5 (+ its bias), once per object; the iterations add the xyz half and apply the ReLU
  c.pconv(point_gemm(r.pf, 384, 192, 192, c.w("feat.conv5.we"), c.w("feat.conv5.bias"), r.e5, 512, 0, 512, rows, ACT_NONE));
}

static void refiner_iterate(Ctx &c, const RefinerBufs &r, int B, int N, int Npad, const float *cloud, const float *rt,
                            const int64_t *obj, float *out_r, float *out_t, double *state, float *rt_next, double *pose_out) {
  const int rows = B * Npad;
  if (point_fused()) {
    PointFeatParams q;
    q.cloud = cloud; q.rt = rt; q.pf = r.pf; q.ld = 384; q.cx1 = 0; q.cx2 = 64;
    q.w1 = c.w("feat.conv1.weight"); q.b1 = c.w("feat.conv1.bias"); q.w2 = c.w("feat.conv2.weight"); q.b2 = c.w("feat.conv2.bias");
    q.B = B; q.N = N; q.Npad = Npad;
    if (c.live()) { const int rc = launch_pointfeat(q, c.st); if (rc != DF_OK) c.err = rc; }
  } else {
    if (c.live()) launch_cloud_conv1(cloud, rt, c.w("feat.conv1.weight"), c.w("feat.conv1.bias"), r.pf, 384, B, N, Npad, c.st);
    c.pconv(point_gemm(r.pf, 384, 0, 64, c.w("feat.conv2.weight"), c.w("feat.conv2.bias"), r.pf, 384, 64, 128, rows, ACT_RELU));
  }
  {
    ConvParams p = point_gemm(r.pf, 384, 0, 192, c.w("feat.conv5.wx"), nullptr, r.x5, 512, 0, 512, rows, ACT_RELU);
    p.res = r.e5; p.res_ld = 512;
    c.pconv(p);
  }
  ConvParams p6 = point_gemm(r.x5, 512, 0, 512, c.w("feat.conv6.weight"), c.w("feat.conv6.bias"), nullptr, 1024, 0, 1024, rows, ACT_RELU);
  p6.rows_per_group = Npad; p6.rows_valid = N; p6.colsum = r.partial;
  c.pconv(p6);
  if (c.live()) launch_colsum_finish(r.partial, r.prow / B, r.apx, B, 1024, N, c.st);
  // FC towers 1024 -> 512 -> 128 for r and t (lib/network.py:191-196): one row per object
  if (c.live()) {
    launch_fc_rows(r.apx, 1024, 0, c.w("fc1.w"), c.w("fc1.bias"), r.f1, 1024, B, 1024, 1024, 1, 1, c.st);
    launch_fc_rows(r.f1, 1024, 512, c.w("fc2.w"), c.w("fc2.bias"), r.f2, 256, B, 512, 128, 2, 1, c.st);
  }
  if (c.live()) {
    launch_refiner_tail(r.f2, c.w("conv3_r.weight"), c.w("conv3_r.bias"), c.w("conv3_t.weight"), c.w("conv3_t.bias"), obj,
                        c.net->num_obj, out_r, out_t, state, rt_next, pose_out, B, c.st);
  }
}

static Net *as_net(df_net *h) { return reinterpret_cast<Net *>(h); }
static const Net *as_net(const df_net *h) { return reinterpret_cast<const Net *>(h); }

static int finish(Ctx &c, const char *what) {
  if (c.err != DF_OK) return c.err;
  return c.dry ? DF_OK : check_launch(what);
}

}  // namespace df

using namespace df;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" df_net *df_posenet_create(int num_points, int num_obj) {
  if (num_points <= 0 || num_obj <= 0) { set_error(DF_ERR_ARG, "posenet_create: bad sizes"); return nullptr; }
  Net *n = new Net();
  n->kind = 0; n->num_points = num_points; n->num_obj = num_obj;
  hipGetDevice(&n->device);
  build_posenet_spec(*n);
  n->loaded.assign(n->spec.size(), 0);
  return reinterpret_cast<df_net *>(n);
}

extern "C" df_net *df_refiner_create(int num_points, int num_obj) {
  if (num_points <= 0 || num_obj <= 0) { set_error(DF_ERR_ARG, "refiner_create: bad sizes"); return nullptr; }
  Net *n = new Net();
  n->kind = 1; n->num_points = num_points; n->num_obj = num_obj;
  hipGetDevice(&n->device);
  build_refiner_spec(*n);
  n->loaded.assign(n->spec.size(), 0);
  return reinterpret_cast<df_net *>(n);
}

extern "C" void df_net_destroy(df_net *h) {
  if (!h) return;
  Net *n = as_net(h);
#ifdef DF_DEV
  split_gemm_invalidate();
#endif
  for (auto &kv : n->buf) hipFree(kv.second);
  if (n->stage) hipFree(n->stage);
  for (auto &kv : n->taps) if (kv.second.buf) hipFree(kv.second.buf);
  for (auto e : n->ev) hipEventDestroy(e);
  delete n;
}

extern "C" int df_net_num_params(const df_net *h) { return h ? (int)as_net(h)->spec.size() : 0; }

extern "C" int df_net_param_info(const df_net *h, int i, char *key_out, int key_cap, int64_t *shape4, int *ndim) {
  if (!h) return set_error(DF_ERR_ARG, "param_info: null handle");
  const Net *n = as_net(h);
  if (i < 0 || i >= (int)n->spec.size()) return set_error(DF_ERR_ARG, "param_info: index out of range");
  const ParamInfo &p = n->spec[i];
  if (key_out && key_cap > 0) { strncpy(key_out, p.key.c_str(), key_cap - 1); key_out[key_cap - 1] = 0; }
  if (shape4) for (int d = 0; d < 4; ++d) shape4[d] = p.shape[d];
  if (ndim) *ndim = p.ndim;
  return DF_OK;
}

extern "C" int df_net_load_param(df_net *h, const char *key, const float *ptr, int64_t numel) {
  if (!h || !key) return set_error(DF_ERR_ARG, "load_param: null handle/key");
#ifdef DF_DEV
  split_gemm_invalidate();
#endif
  return load_param(*as_net(h), key, ptr, numel);
}

extern "C" int df_net_profile(df_net *h, int enable) {
  if (!h) return set_error(DF_ERR_ARG, "profile: null handle");
  Net *n = as_net(h);
  n->profiling = enable != 0;
  n->ev_used = 0;
  n->ev_flops.clear();
  n->ev_bytes.clear();
  n->ev_useful.clear();
  n->ev_desc.clear();
  return DF_OK;
}

// after a stream sync: sum of GEMM launch durations (ms), their algorithmic FLOPs and count since df_net_profile(1)
extern "C" int df_net_profile_read(df_net *h, double *gemm_ms, double *gemm_flops, double *gemm_useful_flops, double *gemm_bytes,
                                   int *launches) {
  if (!h) return set_error(DF_ERR_ARG, "profile_read: null handle");
  Net *n = as_net(h);
  double ms = 0, fl = 0, by = 0, us = 0;
  static const bool verbose = df::dev_getenv("DF_PROFILE_VERBOSE") != nullptr;
  for (size_t i = 0; i + 1 < n->ev_used; i += 2) {
    float t = 0;
    if (hipEventElapsedTime(&t, n->ev[i], n->ev[i + 1]) != hipSuccess) return set_error(DF_ERR_LAUNCH, "profile_read: events not complete");
    ms += t;
    fl += n->ev_flops[i / 2];
    by += n->ev_bytes[i / 2];
    us += n->ev_useful[i / 2];
    if (verbose)
      fprintf(stderr, "[df-gemm] %s  %.1f us  %.1f TFLOP/s\n", n->ev_desc[i / 2].c_str(), t * 1e3, n->ev_flops[i / 2] / t / 1e9);
  }
  if (gemm_ms) *gemm_ms = ms;
  if (gemm_flops) *gemm_flops = fl;
  if (gemm_useful_flops) *gemm_useful_flops = us;
  if (gemm_bytes) *gemm_bytes = by;
  if (launches) *launches = (int)(n->ev_used / 2);
  n->ev_used = 0;
  n->ev_flops.clear();
  n->ev_bytes.clear();
  n->ev_useful.clear();
  n->ev_desc.clear();
  return DF_OK;
}

extern "C" int df_net_debug_taps(df_net *h, int enable) {
  if (!h) return set_error(DF_ERR_ARG, "debug_taps: null handle");
  Net *n = as_net(h);
  n->taps_on = enable != 0;
  if (!enable) {
    for (auto &kv : n->taps) if (kv.second.buf) hipFree(kv.second.buf);
    n->taps.clear();
  }
  return DF_OK;
}

extern "C" int df_net_debug_tap_read(df_net *h, const char *name, float *dst, int64_t cap, int64_t *shape4) {
  if (!h || !name) return set_error(DF_ERR_ARG, "debug_tap_read: null handle / name");
  Net *n = as_net(h);
  auto it = n->taps.find(name);
  if (it == n->taps.end() || !it->second.buf) return set_error(DF_ERR_STATE, "debug_tap_read: no tap named '%s' (arm df_net_debug_taps and run a single-bucket forward)", name);
  const Net::Tap &t = it->second;
  const int64_t numel = t.shape[0] * t.shape[1] * t.shape[2] * t.shape[3];
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
  if (!dst) return DF_OK;          // shape query
  if (cap < numel) return set_error(DF_ERR_ARG, "debug_tap_read: destination holds %lld floats, tap '%s' has %lld", (long long)cap, name, (long long)numel);
  if (hipMemcpy(dst, t.buf, (size_t)numel * sizeof(float), hipMemcpyDefault) != hipSuccess) return set_error(DF_ERR_LAUNCH, "debug_tap_read: copy failed");
  return DF_OK;
}

static int posenet_args_ok(const Net *n, int B, int H, int W) {
  if (!n || n->kind != 0) return set_error(DF_ERR_ARG, "not a PoseNet handle");
  if (B <= 0 || H < 8 || W < 8) return set_error(DF_ERR_ARG, "posenet: need B >= 1 and H, W >= 8 (got %d, %d, %d)", B, H, W);
  // psp_prior_sum keeps per-row / per-column interpolation tables of the 1/8-resolution map in LDS (12.8 KB + 64 (h + w) bytes <= 64 KB);
  // the datasets' crops end at 480 x 640 (datasets/ycb/dataset.py:247-289)
  if (H > DF_MAX_CROP || W > DF_MAX_CROP) return set_error(DF_ERR_ARG, "posenet: crops beyond %d pixels per side are not supported (got %d x %d)", DF_MAX_CROP, H, W);
  return DF_OK;
}

// bucket list of a multi-bucket call (host arrays of nb entries); img may be null (workspace sizing)
static int make_groups(const Net *n, int nb, const int *B, const int *H, const int *W, const float *const *img, std::vector<Grp> &gs) {
  // 4096: far above the 12 x 16 = 192 crop sizes the datasets' 40-pixel snapping can produce (datasets/ycb/dataset.py:247-289)
  if (nb <= 0 || nb > 4096 || !B || !H || !W) return set_error(DF_ERR_ARG, "need 1..4096 buckets with B / H / W arrays (got nb = %d)", nb);
  long tot = 0;
  gs.clear();
  for (int i = 0; i < nb; ++i) {
    const int rc = posenet_args_ok(n, B[i], H[i], W[i]);
    if (rc != DF_OK) return rc;
    if (img && !img[i]) return set_error(DF_ERR_ARG, "bucket %d: null image pointer", i);
    gs.push_back(Grp{B[i], H[i], W[i], img ? img[i] : nullptr});
    tot += B[i];
  }
  if (tot > (1 << 20)) return set_error(DF_ERR_ARG, "too many objects in one call (%ld)", tot);
  return DF_OK;
}

extern "C" size_t df_posenet_workspace_bytes(const df_net *h, int B, int H, int W) {
  if (posenet_args_ok(as_net(h), B, H, W) != DF_OK) return 0;
  Ctx c{const_cast<Net *>(as_net(h)), nullptr, true, nullptr};
  PoseNetOut o{};
  posenet_forward(c, {Grp{B, H, W, nullptr}}, nullptr, nullptr, nullptr, o);
  return c.peak;
}

extern "C" int df_posenet_forward(df_net *h, int B, int H, int W, const float *img, const float *cloud,
                                  const int64_t *choose, const int64_t *obj, float *out_r, float *out_t, float *out_c,
                                  float *emb, void *ws, size_t ws_bytes, df_stream_t stream) {
  int rc = posenet_args_ok(as_net(h), B, H, W);
  if (rc != DF_OK) return rc;
  if ((rc = check_ready(*as_net(h))) != DF_OK) return rc;
  if (!img || !cloud || !choose || !obj || !out_r || !out_t || !out_c || !emb || !ws) return set_error(DF_ERR_ARG, "posenet_forward: null pointer");
  Ctx c{as_net(h), to_stream(stream), false, static_cast<char *>(ws)};
  c.cap = ws_bytes;
  if (df_posenet_workspace_bytes(h, B, H, W) > ws_bytes) return set_error(DF_ERR_WORKSPACE, "posenet_forward: workspace too small");
  PoseNetOut o{out_r, out_t, out_c, emb};
  posenet_forward(c, {Grp{B, H, W, img}}, cloud, choose, obj, o);
  return finish(c, "posenet_forward");
}

// the same forward over buckets of different crop sizes in ONE pass (objects concatenated in bucket order): what the refiner phase of
// tools/train.py:139-145 needs of its frozen estimator for a whole accumulation window
extern "C" size_t df_posenet_multi_workspace_bytes(const df_net *h, int nb, const int *B, const int *H, const int *W) {
  std::vector<Grp> gs;
  if (!h || as_net(h)->kind != 0 || make_groups(as_net(h), nb, B, H, W, nullptr, gs) != DF_OK) return 0;
  Ctx c{const_cast<Net *>(as_net(h)), nullptr, true, nullptr};
  PoseNetOut o{};
  posenet_forward(c, gs, nullptr, nullptr, nullptr, o);
  return c.peak;
}

extern "C" int df_posenet_forward_multi(df_net *h, int nb, const int *B, const int *H, const int *W, const float *const *img, const float *cloud,
                                        const int64_t *choose, const int64_t *obj, float *out_r, float *out_t, float *out_c, float *emb, void *ws,
                                        size_t ws_bytes, df_stream_t stream) {
  if (!h || as_net(h)->kind != 0) return set_error(DF_ERR_ARG, "not a PoseNet handle");
  if (!img) return set_error(DF_ERR_ARG, "posenet_forward_multi: null pointer");
  std::vector<Grp> gs;
  int rc = make_groups(as_net(h), nb, B, H, W, img, gs);
  if (rc != DF_OK) return rc;
  if ((rc = check_ready(*as_net(h))) != DF_OK) return rc;
  if (!cloud || !choose || !obj || !out_r || !out_t || !out_c || !emb || !ws) return set_error(DF_ERR_ARG, "posenet_forward_multi: null pointer");
  if (df_posenet_multi_workspace_bytes(h, nb, B, H, W) > ws_bytes) return set_error(DF_ERR_WORKSPACE, "posenet_forward_multi: workspace too small");
  Ctx c{as_net(h), to_stream(stream), false, static_cast<char *>(ws)};
  c.cap = ws_bytes;
  PoseNetOut o{out_r, out_t, out_c, emb};
  posenet_forward(c, gs, cloud, choose, obj, o);
  return finish(c, "posenet_forward_multi");
}

static void refiner_standalone(Ctx &c, int B, const float *x, const float *emb, const int64_t *obj, float *out_r, float *out_t) {
  const int N = c.net->num_points, Npad = round_up(N, 128);
  float *emb_pm = c.f((size_t)B * Npad * 32);
  RefinerBufs r = refiner_alloc(c, B, N, Npad);
  if (c.live()) launch_emb_to_pm(emb, emb_pm, B, N, Npad, c.st);
  refiner_prepare(c, r, B, Npad, emb_pm);
  refiner_iterate(c, r, B, N, Npad, x, nullptr, obj, out_r, out_t, nullptr, nullptr, nullptr);
}

extern "C" size_t df_refiner_workspace_bytes(const df_net *h, int B) {
  if (!h || as_net(h)->kind != 1 || B <= 0) return 0;
  Ctx c{const_cast<Net *>(as_net(h)), nullptr, true, nullptr};
  refiner_standalone(c, B, nullptr, nullptr, nullptr, nullptr, nullptr);
  return c.peak;
}

extern "C" int df_refiner_forward(df_net *h, int B, const float *x, const float *emb, const int64_t *obj, float *out_r,
                                  float *out_t, void *ws, size_t ws_bytes, df_stream_t stream) {
  if (!h || as_net(h)->kind != 1) return set_error(DF_ERR_ARG, "not a PoseRefineNet handle");
  if (B <= 0) return set_error(DF_ERR_ARG, "refiner_forward: B must be >= 1");
  int rc = check_ready(*as_net(h));
  if (rc != DF_OK) return rc;
  if (!x || !emb || !obj || !out_r || !out_t || !ws) return set_error(DF_ERR_ARG, "refiner_forward: null pointer");
  if (df_refiner_workspace_bytes(h, B) > ws_bytes) return set_error(DF_ERR_WORKSPACE, "refiner_forward: workspace too small");
  Ctx c{as_net(h), to_stream(stream), false, static_cast<char *>(ws)};
  c.cap = ws_bytes;
  refiner_standalone(c, B, x, emb, obj, out_r, out_t);
  return finish(c, "refiner_forward");
}

// PoseNet -> selection -> `iters` refine passes, entirely on the device, for the objects of all buckets (concatenated in bucket order)
static void estimate(Ctx &cp, Ctx &cr, const std::vector<Grp> &gs, const float *cloud, const int64_t *choose, const int64_t *obj, int iters,
                     double *pose_wo, double *pose) {
  const int N = cp.net->num_points, Npad = round_up(N, 128);
  int B = 0;
  for (const Grp &g : gs) B += g.B;
  PoseNetOut o{};
  o.emb = cp.f((size_t)B * 32 * N);
  double *state = reinterpret_cast<double *>(cp.bytes((size_t)B * 7 * sizeof(double)));
  float *rt = cp.f((size_t)B * 12);
  // eval_ycb.py:193-203 reads the r / t heads at the arg-max-confidence point only: confidence tower for all points,
  // r / t towers for that one point (posenet_points, `sel`)
  const SelectOut sel{pose_wo, state, rt};
  posenet_forward(cp, gs, cloud, choose, obj, o, &sel);
  // the refiner context continues in the same workspace
  cr.off = cp.off;
  cr.peak = cp.peak;
  RefinerBufs r = refiner_alloc(cr, B, N, Npad);
  if (iters > 0) refiner_prepare(cr, r, B, Npad, o.emb_pm);
  for (int it = 0; it < iters; ++it)
    refiner_iterate(cr, r, B, N, Npad, cloud, rt, obj, nullptr, nullptr, state, rt, it == iters - 1 ? pose : nullptr);
  if (iters == 0 && cp.live() && pose) hipMemcpyAsync(pose, state, (size_t)B * 7 * sizeof(double), hipMemcpyDeviceToDevice, cp.st);
}

static int estimate_handles_ok(const df_net *pn, const df_net *rf) {
  if (!pn || as_net(pn)->kind != 0) return set_error(DF_ERR_ARG, "estimate_poses: not a PoseNet handle");
  if (!rf || as_net(rf)->kind != 1) return set_error(DF_ERR_ARG, "estimate_poses: not a PoseRefineNet handle");
  if (as_net(rf)->num_points != as_net(pn)->num_points || as_net(rf)->num_obj != as_net(pn)->num_obj)
    return set_error(DF_ERR_ARG, "estimate_poses: estimator / refiner disagree on num_points or num_obj");
  return DF_OK;
}

extern "C" size_t df_estimate_multi_workspace_bytes(const df_net *pn, const df_net *rf, int nb, const int *B, const int *H, const int *W) {
  std::vector<Grp> gs;
  if (estimate_handles_ok(pn, rf) != DF_OK || make_groups(as_net(pn), nb, B, H, W, nullptr, gs) != DF_OK) return 0;
  Ctx cp{const_cast<Net *>(as_net(pn)), nullptr, true, nullptr}, cr{const_cast<Net *>(as_net(rf)), nullptr, true, nullptr};
  estimate(cp, cr, gs, nullptr, nullptr, nullptr, 1, nullptr, nullptr);
  return cr.peak;
}

extern "C" int df_estimate_poses_multi(df_net *pn, df_net *rf, int nb, const int *B, const int *H, const int *W, const float *const *img,
                                       const float *cloud, const int64_t *choose, const int64_t *obj, int iters, double *pose_wo,
                                       double *pose, void *ws, size_t ws_bytes, df_stream_t stream) {
  int rc = estimate_handles_ok(pn, rf);
  if (rc != DF_OK) return rc;
  std::vector<Grp> gs;
  if (!img) return set_error(DF_ERR_ARG, "estimate_poses: null pointer");
  if ((rc = make_groups(as_net(pn), nb, B, H, W, img, gs)) != DF_OK) return rc;
  if (iters < 0) return set_error(DF_ERR_ARG, "estimate_poses: iters < 0");
  if ((rc = check_ready(*as_net(pn))) != DF_OK || (rc = check_ready(*as_net(rf))) != DF_OK) return rc;
  if (!cloud || !choose || !obj || !pose || !ws) return set_error(DF_ERR_ARG, "estimate_poses: null pointer");
  if (df_estimate_multi_workspace_bytes(pn, rf, nb, B, H, W) > ws_bytes) return set_error(DF_ERR_WORKSPACE, "estimate_poses: workspace too small");
  Ctx cp{as_net(pn), to_stream(stream), false, static_cast<char *>(ws)}, cr{as_net(rf), to_stream(stream), false, static_cast<char *>(ws)};
  cp.cap = cr.cap = ws_bytes;
  estimate(cp, cr, gs, cloud, choose, obj, iters, pose_wo, pose);
  if (cp.err != DF_OK) return cp.err;
  return finish(cr, "estimate_poses");
}

extern "C" size_t df_estimate_workspace_bytes(const df_net *pn, const df_net *rf, int B, int H, int W) {
  return df_estimate_multi_workspace_bytes(pn, rf, 1, &B, &H, &W);
}

extern "C" int df_estimate_poses(df_net *pn, df_net *rf, int B, int H, int W, const float *img, const float *cloud,
                                 const int64_t *choose, const int64_t *obj, int iters, double *pose_wo, double *pose,
                                 void *ws, size_t ws_bytes, df_stream_t stream) {
  return df_estimate_poses_multi(pn, rf, 1, &B, &H, &W, &img, cloud, choose, obj, iters, pose_wo, pose, ws, ws_bytes, stream);
}

static int conv_desc_to_params(const df_conv_desc *d, ConvParams &p, const char *what);

static thread_local int t_last_splitk = 1;
extern "C" int df_conv_last_splitk(void) { return t_last_splitk; }

extern "C" int df_conv2d_nhwc(const df_conv_desc *d, df_stream_t stream) {
  if (!d) return set_error(DF_ERR_ARG, "conv2d_nhwc: null descriptor");
  if (d->KH != d->KW) return set_error(DF_ERR_ARG, "conv2d_nhwc: square kernels only");
  if (d->act == ACT_PRELU && !d->prelu) return set_error(DF_ERR_ARG, "conv2d_nhwc: PReLU needs a slope");
  ConvParams p;
  p.in = d->in; p.wgt = d->wgt; p.bias = d->bias; p.res = d->res; p.prelu = d->prelu; p.out = d->out;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_ld = d->in_ld; p.in_coff = d->in_coff;
  p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout; p.out_ld = d->out_ld; p.out_coff = d->out_coff;
  p.res_ld = d->res_ld; p.res_coff = d->res_coff;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.act = d->act;
  if (p.OH != conv_out(p.H, p.KH, p.stride, p.pad, p.dil) || p.OW != conv_out(p.W, p.KW, p.stride, p.pad, p.dil))
    return set_error(DF_ERR_ARG, "conv2d_nhwc: OH/OW do not match the convolution geometry");
  p.splitk_ws = static_cast<float *>(d->splitk_ws); p.splitk_ws_bytes = d->splitk_ws ? d->splitk_ws_bytes : 0;
  return launch_conv(p, to_stream(stream), &t_last_splitk);
}

// buckets of a multi-bucket convolution / weight-gradient call: pixel rows concatenated in bucket order in both operands
static int make_segs(const df_conv_desc *d, int nb, const int *B, const int *H, const int *W, std::vector<WgradSeg> &segs, const char *what) {
  if (!d) return set_error(DF_ERR_ARG, "%s: null descriptor", what);
  if (nb <= 0 || nb > 4096 || !B || !H || !W) return set_error(DF_ERR_ARG, "%s: need 1..4096 buckets with B / H / W arrays", what);
  if (d->KH != d->KW) return set_error(DF_ERR_ARG, "%s: square kernels only", what);
  long in_row = 0, out_row = 0;
  for (int i = 0; i < nb; ++i) {
    if (B[i] <= 0 || H[i] <= 0 || W[i] <= 0) return set_error(DF_ERR_ARG, "%s: bucket %d is empty", what, i);
    const int OH = conv_out(H[i], d->KH, d->stride, d->pad, d->dil), OW = conv_out(W[i], d->KW, d->stride, d->pad, d->dil);
    if (OH <= 0 || OW <= 0) return set_error(DF_ERR_ARG, "%s: bucket %d: the kernel does not fit the map", what, i);
    segs.push_back(WgradSeg{B[i], H[i], W[i], OH, OW, in_row, out_row});
    in_row += (long)B[i] * H[i] * W[i];
    out_row += (long)B[i] * OH * OW;
  }
  return DF_OK;
}

extern "C" int df_conv2d_nhwc_multi(const df_conv_desc *d, int nb, const int *B, const int *H, const int *W, df_stream_t stream) {
  std::vector<WgradSeg> segs;
  int rc = make_segs(d, nb, B, H, W, segs, "conv2d_nhwc_multi");
  if (rc != DF_OK) return rc;
  if (d->act == ACT_PRELU && !d->prelu) return set_error(DF_ERR_ARG, "conv2d_nhwc_multi: PReLU needs a slope");
  ConvParams p;
  p.in = d->in; p.wgt = d->wgt; p.bias = d->bias; p.res = d->res; p.prelu = d->prelu; p.out = d->out;
  p.Cin = d->Cin; p.in_ld = d->in_ld; p.in_coff = d->in_coff; p.Cout = d->Cout; p.out_ld = d->out_ld; p.out_coff = d->out_coff;
  p.res_ld = d->res_ld; p.res_coff = d->res_coff;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.act = d->act;
  return launch_conv_multi(p, nb, segs.data(), to_stream(stream));
}

extern "C" size_t df_conv2d_wgrad_multi_workspace_bytes(const df_conv_desc *d, int nb, const int *B, const int *H, const int *W) {
  std::vector<WgradSeg> segs;
  if (make_segs(d, nb, B, H, W, segs, "conv2d_wgrad_multi_workspace_bytes") != DF_OK) return 0;
  ConvParams p;
  p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
  return wgrad_multi_workspace_bytes(p, nb, segs.data());
}

extern "C" int df_conv2d_wgrad_nhwc_multi(const df_conv_desc *d, int nb, const int *B, const int *H, const int *W, const float *dy, float *dw, float *db,
                                          void *ws, size_t ws_bytes, df_stream_t stream) {
  std::vector<WgradSeg> segs;
  int rc = make_segs(d, nb, B, H, W, segs, "conv2d_wgrad_multi");
  if (rc != DF_OK) return rc;
  if (!dy || !dw || !d->in) return set_error(DF_ERR_ARG, "conv2d_wgrad_multi: null pointer");
  ConvParams p;
  p.in = d->in; p.out = const_cast<float *>(dy);
  p.Cin = d->Cin; p.in_ld = d->in_ld; p.in_coff = d->in_coff; p.Cout = d->Cout; p.out_ld = d->out_ld; p.out_coff = d->out_coff;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  return launch_wgrad_multi(p, nb, segs.data(), dw, db, ws, ws_bytes, to_stream(stream));
}

// 3x3 stride-1 pad=dil convolution through the Winograd F(2x2,3x3) / F(4x4,3x3) domain (wino.hip): weight transform, input
// transform, 16 / 36 batched GEMMs, output transform (+ bias, residual, ReLU).  scratch holds U | V | M.
static int wino_desc_ok(const df_conv_desc *d, const char *what) {
  if (!d) return set_error(DF_ERR_ARG, "%s: null descriptor", what);
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != d->dil || d->dil < 1)
    return set_error(DF_ERR_ARG, "%s: needs a 3x3 kernel, stride 1, pad == dil", what);
  if (d->Cin % 4 || d->Cout % 4 || d->in_ld % 4 || d->in_coff % 4 || d->out_ld % 4 || d->out_coff % 4 || d->B <= 0 || d->H <= 0 || d->W <= 0)
    return set_error(DF_ERR_ARG, "%s: channel counts / strides must be multiples of 4", what);
  if (d->OH != d->H || d->OW != d->W) return set_error(DF_ERR_ARG, "%s: OH/OW must equal H/W", what);
  if (d->act != ACT_NONE && d->act != ACT_RELU) return set_error(DF_ERR_ARG, "%s: activation must be none or ReLU", what);
  return DF_OK;
}

extern "C" size_t df_conv3x3_winograd_tile_scratch_bytes(const df_conv_desc *d, int tile) {
  if (wino_desc_ok(d, "conv3x3_winograd_scratch_bytes") != DF_OK) return 0;
  if (tile != 2 && tile != 4) { set_error(DF_ERR_ARG, "conv3x3_winograd: tile must be 2 or 4"); return 0; }
  const WinoGeom g = wino_geom(d->B, d->H, d->W, d->dil, tile);
  const size_t nz = (size_t)(tile + 2) * (tile + 2);
  return (nz * d->Cout * d->Cin + nz * g.T * d->Cin + nz * g.T * d->Cout) * sizeof(float);
}

extern "C" int df_conv3x3_winograd_tile_nhwc(const df_conv_desc *d, int tile, void *scratch, size_t scratch_bytes, df_stream_t stream) {
  int rc = wino_desc_ok(d, "conv3x3_winograd_nhwc");
  if (rc != DF_OK) return rc;
  if (tile != 2 && tile != 4) return set_error(DF_ERR_ARG, "conv3x3_winograd: tile must be 2 or 4");
  if (!d->in || !d->wgt || !d->out || !scratch) return set_error(DF_ERR_ARG, "conv3x3_winograd_nhwc: null pointer");
  if (scratch_bytes < df_conv3x3_winograd_tile_scratch_bytes(d, tile)) return set_error(DF_ERR_WORKSPACE, "conv3x3_winograd_nhwc: scratch too small");
  const WinoGeom g = wino_geom(d->B, d->H, d->W, d->dil, tile);
  const int nz = (tile + 2) * (tile + 2);
  hipStream_t st = to_stream(stream);
  float *U = static_cast<float *>(scratch), *V = U + (size_t)nz * d->Cout * d->Cin, *M = V + (size_t)nz * g.T * d->Cin;
  launch_wino_weight(d->wgt, U, d->Cout, d->Cin, st, tile);
  launch_wino_input(d->in, d->in_ld, d->in_coff, V, d->B, d->H, d->W, d->Cin, d->dil, st, 0, 0, tile);
  ConvParams p;
  p.in = V; p.wgt = U; p.out = M;
  p.B = (int)g.T; p.Cin = d->Cin; p.in_ld = d->Cin; p.Cout = d->Cout; p.out_ld = d->Cout;
  p.zcount = nz; p.z_in_coff = g.T * d->Cin; p.z_wgt = (long)d->Cout * d->Cin; p.z_out_coff = g.T * d->Cout;
  rc = launch_conv(p, st);
  if (rc != DF_OK) return rc;
  launch_wino_output(M, d->out, d->out_ld, d->out_coff, d->bias, d->res, d->res_ld, d->res_coff, d->act, d->B, d->H, d->W, d->Cout, d->dil, st, 0, 0,
                     tile);
  return check_launch("conv3x3_winograd_nhwc");
}

extern "C" int df_wino_route(int H, int W, int dil, int Cin, int Cout) { return wino_route(H, W, dil, Cin, Cout); }

extern "C" size_t df_conv3x3_winograd_scratch_bytes(const df_conv_desc *d) { return df_conv3x3_winograd_tile_scratch_bytes(d, 2); }

extern "C" int df_conv3x3_winograd_nhwc(const df_conv_desc *d, void *scratch, size_t scratch_bytes, df_stream_t stream) {
  return df_conv3x3_winograd_tile_nhwc(d, 2, scratch, scratch_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// training building blocks: data gradient and weight gradient of df_conv2d_nhwc
// ------------------------------------------------------------------------------------------------
namespace df {
// wt[c][ky][kx][n] = w[n][KH-1-ky][KW-1-kx][c]   (the forward conv's weights as seen by its data gradient)
__global__ void flip_transpose_kernel(const float *__restrict__ w, float *__restrict__ wt, int O, int T, int I, int KH, int KW) {
  const long total = (long)O * T * I;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i % O);
    const long r = i / O;
    const int t = (int)(r % T);
    const long c = r / T;
    const int ky = t / KW, kx = t - ky * KW;
    const int tf = (KH - 1 - ky) * KW + (KW - 1 - kx);
    wt[i] = w[((size_t)n * T + tf) * I + c];
  }
}
}  // namespace df

static int conv_desc_to_params(const df_conv_desc *d, ConvParams &p, const char *what) {
  if (!d) return set_error(DF_ERR_ARG, "%s: null descriptor", what);
  if (d->KH != d->KW) return set_error(DF_ERR_ARG, "%s: square kernels only", what);
  p.in = d->in; p.wgt = d->wgt; p.bias = d->bias; p.res = d->res; p.prelu = d->prelu; p.out = d->out;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_ld = d->in_ld; p.in_coff = d->in_coff;
  p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout; p.out_ld = d->out_ld; p.out_coff = d->out_coff;
  p.res_ld = d->res_ld; p.res_coff = d->res_coff;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.act = d->act;
  if (p.OH != conv_out(p.H, p.KH, p.stride, p.pad, p.dil) || p.OW != conv_out(p.W, p.KW, p.stride, p.pad, p.dil))
    return set_error(DF_ERR_ARG, "%s: OH/OW do not match the convolution geometry", what);
  return DF_OK;
}

extern "C" int df_conv2d_dgrad_nhwc(const df_conv_desc *d, const float *dy, float *dx, float *w_scratch, int accumulate,
                                    df_stream_t stream) {
  ConvParams f;
  int rc = conv_desc_to_params(d, f, "conv2d_dgrad");
  if (rc != DF_OK) return rc;
  if (!dy || !dx || !w_scratch || !f.wgt) return set_error(DF_ERR_ARG, "conv2d_dgrad: null pointer");
  if (f.Cout % 4) return set_error(DF_ERR_ARG, "conv2d_dgrad: Cout must be a multiple of 4");
  hipStream_t st = to_stream(stream);
  const int T = f.KH * f.KW;
  hipLaunchKernelGGL(flip_transpose_kernel, dim3(256), dim3(256), 0, st, f.wgt, w_scratch, f.Cout, T, f.Cin, f.KH, f.KW);
  ConvParams q;
  q.in = dy; q.B = f.B; q.H = f.OH; q.W = f.OW; q.Cin = f.Cout; q.in_ld = f.out_ld; q.in_coff = f.out_coff;
  q.wgt = w_scratch;
  q.out = dx; q.OH = f.H; q.OW = f.W; q.Cout = f.Cin; q.out_ld = f.in_ld; q.out_coff = f.in_coff;
  q.KH = f.KH; q.KW = f.KW; q.stride = 1; q.up = f.stride; q.dil = f.dil; q.pad = f.dil * (f.KH - 1) - f.pad;
  if (q.pad < 0) return set_error(DF_ERR_ARG, "conv2d_dgrad: padding larger than the kernel reach is not supported");
  if (accumulate) { q.res = dx; q.res_ld = f.in_ld; q.res_coff = f.in_coff; }
  q.splitk_ws = static_cast<float *>(d->splitk_ws); q.splitk_ws_bytes = d->splitk_ws ? d->splitk_ws_bytes : 0;
  return launch_conv(q, st, &t_last_splitk);
}

extern "C" size_t df_conv2d_wgrad_workspace_bytes(const df_conv_desc *d) {
  ConvParams f;
  if (conv_desc_to_params(d, f, "conv2d_wgrad_workspace_bytes") != DF_OK) return 0;
  return wgrad_workspace_bytes(f);
}

extern "C" int df_conv2d_wgrad_nhwc(const df_conv_desc *d, const float *dy, float *dw, float *db, void *ws, size_t ws_bytes, df_stream_t stream) {
  ConvParams f;
  int rc = conv_desc_to_params(d, f, "conv2d_wgrad");
  if (rc != DF_OK) return rc;
  if (!dy || !dw || !f.in) return set_error(DF_ERR_ARG, "conv2d_wgrad: null pointer");
  f.out = const_cast<float *>(dy);
  return launch_wgrad(f, dw, db, ws, ws_bytes, to_stream(stream));
}
