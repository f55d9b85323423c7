// Sanitizer dry-run of the engine's HOST logic through the C ABI (include/neurons_amd.h), linked against tests/sanitize/hip_stub.cpp
// instead of the HIP runtime and built with -fsanitize=address,undefined (tests/test_sanitize_host.py).  What runs for real: state-dict
// loading, every weight conversion (tap-inner / GEGLU interleave / LayerNorm fold / stacked time-embedding projections), the two-pass
// planner with its first-fit arena, the context / persistent region split, launch-plan construction (the per-shape kernel choices, the
// split-K workspace sizing, fused-kernel stream packing), hipGraph capture bookkeeping and the 64-slot LRU, manifest export / parse /
// import, reload invalidation, host-weight release, error paths.  Kernel launches are no-ops.
//   usage: planner_dryrun <schema-file>      schema lines:  N <name> <kind> <cfg ints...>   /   T <key> <ndim> <dims...>
#include "../../include/neurons_amd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

extern "C" long nr_stub_live_allocs(void);
extern "C" long nr_stub_live_graphs(void);
extern "C" long nr_stub_captures(void);
extern "C" long nr_stub_launches(void);

struct Net { nr_net_config cfg; std::vector<std::pair<std::string, std::vector<int64_t>>> tensors; };

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s (%s)\n", __FILE__, __LINE__, msg, nr_last_error()); exit(2); } } while (0)
#define OK(call) CHECK((call) == NR_OK, #call)

static void load_all(nr_net* h, const Net& n, unsigned seed) {
  for (auto& t : n.tensors) {
    int64_t numel = 1;
    for (auto d : t.second) numel *= d;
    std::vector<float> data((size_t)numel);
    unsigned s = seed * 2654435761u + (unsigned)std::hash<std::string>()(t.first);
    const bool vec = t.second.size() == 1;
    for (auto& v : data) { s = s * 1664525u + 1013904223u; const float u = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; v = vec ? (t.first.find("weight") != std::string::npos ? 1.0f + 0.1f * u : 0.05f * u) : 0.1f * u; }
    OK(nr_net_load_tensor(h, t.first.c_str(), data.data(), t.second.data(), (int32_t)t.second.size()));
  }
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: planner_dryrun <schema-file>\n"); return 1; }
  FILE* f = fopen(argv[1], "r");
  CHECK(f, "schema file");
  std::map<std::string, Net> nets;
  char line[4096], name[256];
  std::string cur;
  while (fgets(line, sizeof(line), f)) {
    if (line[0] == 'N') {
      std::vector<long> v;
      char* tok = strtok(line + 2, " \n");
      strncpy(name, tok, sizeof(name) - 1);
      while ((tok = strtok(nullptr, " \n"))) v.push_back(atol(tok));
      cur = name;
      Net& n = nets[cur];
      std::memset(&n.cfg, 0, sizeof(n.cfg));
      int32_t* ci = reinterpret_cast<int32_t*>(&n.cfg);
      CHECK(v.size() * sizeof(int32_t) == sizeof(nr_net_config), "config width");   // norm_eps travels as its bit pattern
      for (size_t i = 0; i < v.size(); ++i) ci[i] = (int32_t)v[i];
    } else if (line[0] == 'T') {
      char* tok = strtok(line + 2, " \n");
      std::string key = tok;
      const int nd = atoi(strtok(nullptr, " \n"));
      std::vector<int64_t> shape;
      for (int i = 0; i < nd; ++i) shape.push_back(atoll(strtok(nullptr, " \n")));
      nets[cur].tensors.emplace_back(key, shape);
    }
  }
  fclose(f);
  const float ts[64] = {500.f, 500.f, 480.f, 480.f};
  std::vector<float> io((size_t)64 << 20);          // one big host block standing in for every "device" I/O tensor
  float* sample = io.data();
  float* ctx = io.data() + (8 << 20);
  float* out = io.data() + (16 << 20);
  float* cond = io.data() + (24 << 20);
  float* mask = io.data() + (28 << 20);

  // ---------------- tiny U-Net + SparseCtrl: plan / re-plan / eager + graph forwards / grouped schedule / LRU churn ----------------
  {
    const Net& nu = nets.at("tiny_unet");
    const Net& nc = nets.at("tiny_ctrl");
    nr_net *u = nullptr, *c = nullptr;
    OK(nr_net_create(&nu.cfg, &u));
    OK(nr_net_create(&nc.cfg, &c));
    CHECK(nr_net_plan(u, 2, 8, 8, 8, 77) == NR_ERR_MISSING_WEIGHT, "plan without weights must fail with MISSING_WEIGHT");
    load_all(u, nu, 1);
    load_all(c, nc, 2);
    CHECK(nr_net_plan(u, 2, 8, 7, 8, 77) == NR_ERR_ARG, "latent size not a multiple of 8 must be rejected");
    CHECK(nr_net_plan(u, 65, 8, 8, 8, 77) == NR_ERR_ARG, "batch beyond NR_MAX_BATCH must be rejected");
    OK(nr_net_plan(u, 2, 8, 8, 8, 77));
    OK(nr_net_plan(c, 2, 8, 8, 8, 77));
    const int nres = nr_net_num_residuals(c);
    CHECK(nres == 12 && nr_net_num_residuals(u) == 12, "residual count");
    std::vector<void*> res(nres + 1);
    std::vector<std::vector<unsigned short>> resbuf(nres + 1);
    for (int i = 0; i <= nres; ++i) {
      int32_t C, hh, ww;
      OK(nr_net_residual_shape(c, i, &C, &hh, &ww));
      resbuf[i].resize((size_t)8 * 2 * 8 * hh * ww * C);            // room for a group of 4 steps
      res[i] = resbuf[i].data();
    }
    for (int graph = 0; graph < 2; ++graph) {
      OK(nr_net_set_graph(u, graph));
      OK(nr_net_set_graph(c, graph));
      OK(nr_sparsectrl_forward(c, nullptr, sample, ts, ctx, 77, cond, mask, 1, 1.0f, res.data(), res[nres]));
      OK(nr_unet3d_forward(u, nullptr, sample, ts, ctx, 77, (const void* const*)res.data(), res[nres], out));
      OK(nr_unet3d_forward(u, nullptr, sample, ts, ctx, 77, nullptr, nullptr, out));
      OK(nr_denoise_step_forward(u, c, nullptr, sample, ts, ctx, 77, cond, mask, 1, 1.0f, res.data(), res[nres], out, ts));
      OK(nr_denoise_step_forward(u, c, nullptr, sample, ts, ctx, 77, cond, mask, 1, 1.0f, res.data(), res[nres], out, nullptr));
    }
    CHECK(nr_unet3d_forward(u, nullptr, sample, ts, ctx, 76, nullptr, nullptr, out) == NR_ERR_ARG, "wrong ctx_len");
    // graph-slot LRU: 80 different output pointers -> 80 captures of the segments that see `out`, at most 64 live per segment
    for (int i = 0; i < 80; ++i) OK(nr_unet3d_forward(u, nullptr, sample, ts, ctx, 77, nullptr, nullptr, out + 16 * i));
    CHECK(nr_stub_live_graphs() <= 3 * 64 + 3 * 64, "graph cache grew beyond its slots");
    // grouped schedule: SparseCtrl planned for 4 x the CFG batch, two slots, U-Net consuming slices
    OK(nr_net_plan(c, 8, 8, 8, 8, 77));
    for (int slot = 0; slot < 2; ++slot)
      OK(nr_sparsectrl_forward_async(c, nullptr, ts, ctx, 77, cond, mask, 1, 1.0f, res.data(), res[nres], slot));
    for (int stp = 0; stp < 4; ++stp) OK(nr_unet3d_forward_after(u, c, stp & 1, nullptr, sample, ts, ctx, 77, (const void* const*)res.data(), res[nres], out));
    CHECK(nr_sparsectrl_forward_async(c, nullptr, ts, ctx, 77, cond, mask, 1, 1.0f, res.data(), res[nres], 2) == NR_ERR_ARG, "slot range");
    // profile + launch-plan introspection
    nr_profile prof;
    OK(nr_net_profile_last(u, nullptr, &prof));
    CHECK(nr_net_num_ops(u) > 100 && strlen(nr_net_op_desc(u, nr_net_num_ops(u) / 2)) < 200, "op descriptions");
    // debug mode (no buffer reuse) re-plans and keeps taps
    OK(nr_net_set_debug(u, 1));
    OK(nr_net_plan(u, 2, 8, 8, 8, 77));
    OK(nr_unet3d_forward(u, nullptr, sample, ts, ctx, 77, nullptr, nullptr, out));
    CHECK(nr_net_num_taps(u) > 40, "taps");
    std::vector<float> tapbuf(1 << 22);
    int32_t rows, C;
    OK(nr_net_read_tap(u, 3, tapbuf.data(), (int64_t)tapbuf.size(), &rows, &C));
    CHECK(nr_net_read_tap(u, 3, tapbuf.data(), 4, &rows, &C) == NR_ERR_ARG, "short tap buffer");
    OK(nr_net_set_debug(u, 0));
    // another shape, deterministic-batch mode, fp8 flag: all invalidate the plan
    OK(nr_net_set_deterministic_batch(u, 1));
    OK(nr_net_plan(u, 4, 16, 16, 8, 77));
    OK(nr_unet3d_forward(u, nullptr, sample, ts, ctx, 77, nullptr, nullptr, out));
    OK(nr_net_set_deterministic_batch(u, 0));
    OK(nr_net_set_attention_fp8(u, 1));
    CHECK(nr_unet3d_forward(u, nullptr, sample, ts, ctx, 77, nullptr, nullptr, out) == NR_ERR_STATE, "forward after a plan-invalidating switch");
    OK(nr_net_plan(u, 2, 8, 8, 8, 77));
    OK(nr_net_set_attention_fp8(u, 0));
    OK(nr_net_plan(u, 2, 8, 8, 8, 77));
    // reload of one tensor invalidates exactly its converted copies; the next plan rebuilds them
    load_all(u, Net{nu.cfg, {nu.tensors[5], nu.tensors[nu.tensors.size() / 2]}}, 9);
    CHECK(nr_unet3d_forward(u, nullptr, sample, ts, ctx, 77, nullptr, nullptr, out) == NR_ERR_STATE, "forward after reload needs a plan");
    OK(nr_net_plan(u, 2, 8, 8, 8, 77));
    // export -> import into a fresh handle -> same plan without host weights
    int64_t arena = 0;
    const int64_t mlen = nr_net_export_manifest(u, nullptr, 0, &arena);
    CHECK(mlen > 0 && arena > 0, "manifest size");
    std::vector<char> man((size_t)mlen);
    CHECK(nr_net_export_manifest(u, man.data(), mlen, &arena) == mlen, "manifest");
    std::vector<char> blob((size_t)arena);
    CHECK(nr_net_export_weights(u, nullptr, blob.data(), arena - 1) == NR_ERR_ARG, "short export buffer");
    OK(nr_net_export_weights(u, nullptr, blob.data(), arena));
    nr_net* v = nullptr;
    OK(nr_net_create(&nu.cfg, &v));
    // damaged manifests leave the handle importable: truncated line, offset beyond the arena, wrong kind, negative dims, garbage
    {
      std::string bad(man.data(), (size_t)mlen);
      const size_t p = bad.find("\nD ");
      std::string b1 = bad.substr(0, p + 3) + "x 99999999999 12\n";
      CHECK(nr_net_import_weights(v, nullptr, b1.data(), (int64_t)b1.size(), blob.data(), arena) == NR_ERR_ARG, "offset beyond arena");
      std::string b2 = "NRW1 1\n" + bad.substr(bad.find('\n') + 1);
      CHECK(nr_net_import_weights(v, nullptr, b2.data(), (int64_t)b2.size(), blob.data(), arena) == NR_ERR_ARG, "wrong kind");
      std::string b3 = bad.substr(0, bad.find('\n') + 1) + "H k 2 -4 3\n";
      CHECK(nr_net_import_weights(v, nullptr, b3.data(), (int64_t)b3.size(), blob.data(), arena) == NR_ERR_ARG, "negative dim");
      std::string b4 = bad.substr(0, bad.find('\n') + 1) + "Q what is this\n";
      CHECK(nr_net_import_weights(v, nullptr, b4.data(), (int64_t)b4.size(), blob.data(), arena) == NR_ERR_ARG, "unknown record");
      std::string b5 = bad.substr(0, bad.find('\n') + 1) + "H k 9 1 1 1 1 1 1 1 1 1\n";
      CHECK(nr_net_import_weights(v, nullptr, b5.data(), (int64_t)b5.size(), blob.data(), arena) == NR_ERR_ARG, "too many dims");
      CHECK(nr_net_import_weights(v, nullptr, "garbage", 7, blob.data(), arena) == NR_ERR_ARG, "garbage header");
      CHECK(nr_net_import_weights(v, nullptr, man.data(), mlen, blob.data(), 0) == NR_ERR_ARG, "empty arena");
    }
    OK(nr_net_import_weights(v, nullptr, man.data(), mlen, blob.data(), arena));
    CHECK(nr_net_import_weights(v, nullptr, man.data(), mlen, blob.data(), arena) == NR_ERR_STATE, "second import");
    OK(nr_net_plan(v, 2, 8, 8, 8, 77));
    OK(nr_unet3d_forward(v, nullptr, sample, ts, ctx, 77, nullptr, nullptr, out));
    // a shape the exporter never planned needs conversions the importer cannot make: a specific error, not a crash
    const nr_status st = nr_net_plan(v, 2, 8, 16, 16, 77);
    CHECK(st == NR_OK || st == NR_ERR_STATE, "re-plan on an imported handle");
    // release host copies, plan the same shape again (conversions cached), then a new shape may or may not need host data
    OK(nr_net_release_host_weights(u));
    OK(nr_net_plan(u, 2, 8, 8, 8, 77));
    OK(nr_unet3d_forward(u, nullptr, sample, ts, ctx, 77, nullptr, nullptr, out));
    nr_net_destroy(v);
    nr_net_destroy(u);
    nr_net_destroy(c);
  }
  // ---------------- full-width level 0 (C = 320): row-panel / fused-kernel eligibility, stream packing, leaf handles ----------------
  for (const char* nm : {"leaf_temporal", "leaf_transformer"}) {
    const Net& n = nets.at(nm);
    nr_net* h = nullptr;
    OK(nr_net_create(&n.cfg, &h));
    load_all(h, n, 5);
    const bool tr = std::string(nm) == "leaf_transformer";
    OK(nr_net_plan(h, 1, tr ? 2 : 16, tr ? 48 : 16, tr ? 48 : 16, tr ? 77 : 0));
    OK(nr_leaf_forward(h, nullptr, sample, tr ? ctx : nullptr, tr ? 77 : 0, out));
    bool fused = false;
    for (int i = 0; i < nr_net_num_ops(h); ++i) fused = fused || strstr(nr_net_op_desc(h, i), "ff_fused") != nullptr;
    CHECK(fused, "C = 320 at >= 4096 rows must plan the fused FeedForward");
    OK(nr_net_plan(h, 1, tr ? 2 : 16, 8, 8, tr ? 77 : 0));            // below the gate: unfused sequence, same weights
    OK(nr_leaf_forward(h, nullptr, sample, tr ? ctx : nullptr, tr ? 77 : 0, out));
    OK(nr_net_plan(h, 2, tr ? 2 : 16, tr ? 48 : 16, tr ? 48 : 16, tr ? 77 : 0));
    OK(nr_leaf_forward(h, nullptr, sample, tr ? ctx : nullptr, tr ? 77 : 0, out));
    nr_net_destroy(h);
  }
  // ---------------- C = 640 transformer at <= 512 rows: fragment-major weight copies of the panel-resident small-M kernel (smallm.hip) ----------------
  if (nets.count("leaf_transformer640")) {
    const Net& n = nets.at("leaf_transformer640");
    long long bytes_on = 0, bytes_off = 0, bytes_reloaded = 0;
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) setenv("NR_SMALLM", "0", 1);
      nr_net* h = nullptr;
      OK(nr_net_create(&n.cfg, &h));
      load_all(h, n, 9);
      OK(nr_net_plan(h, 2, 2, 8, 8, 77));                             // 256 rows, K = 640: eligible
      OK(nr_leaf_forward(h, nullptr, sample, ctx, 77, out));
      (pass == 0 ? bytes_on : bytes_off) = nr_net_weight_bytes(h);
      if (pass == 0) {
        // reloading a tensor drops the converted weights derived from it -- the row-major matrix AND its fragment-major copy -- and the next plan
        // rebuilds both: same total, nothing leaked
        const long live_before = nr_stub_live_allocs();
        for (auto& t : n.tensors)
          if (t.first.find("attn1.to_out.0.weight") != std::string::npos) {
            int64_t numel = 1;
            for (auto d : t.second) numel *= d;
            std::vector<float> data((size_t)numel, 0.01f);
            OK(nr_net_load_tensor(h, t.first.c_str(), data.data(), t.second.data(), (int32_t)t.second.size()));
          }
        CHECK(nr_leaf_forward(h, nullptr, sample, ctx, 77, out) == NR_ERR_STATE, "forward after a reload must ask for a new plan");
        OK(nr_net_plan(h, 2, 2, 8, 8, 77));
        OK(nr_leaf_forward(h, nullptr, sample, ctx, 77, out));
        bytes_reloaded = nr_net_weight_bytes(h);
        CHECK(nr_stub_live_allocs() == live_before, "reload + re-plan must not leak converted weights");
        OK(nr_net_plan(h, 2, 2, 32, 32, 77));                         // 4096 rows: same weights, no new copies
        OK(nr_leaf_forward(h, nullptr, sample, ctx, 77, out));
        CHECK(nr_net_weight_bytes(h) >= bytes_reloaded, "weights of another shape only add");
      }
      nr_net_destroy(h);
    }
    unsetenv("NR_SMALLM");
    CHECK(bytes_on > bytes_off, "eligible Linears must hold a fragment-major copy of their weights");
    CHECK(bytes_reloaded == bytes_on, "reload must rebuild exactly the dropped copies");
    printf("fragment-major copies: %lld bytes on top of %lld\n", bytes_on - bytes_off, bytes_off);
  }
  // ---------------- C = 640 temporal module, 16 frames: the attention head kernel's planner logic (tattnw.hip, round 6): the fragment-major weight
  // stream and the epilogue table are built once, the LayerNorm-folded [3C][C] matrix that feeds the stream is dropped again, a reloaded to_q
  // drops stream + table + fold vectors, the next plan rebuilds exactly them, nothing leaks; F = 8 (not eligible) plans the three-launch sequence ----------------
  if (nets.count("leaf_temporal640")) {
    const Net& n = nets.at("leaf_temporal640");
    nr_net* h = nullptr;
    OK(nr_net_create(&n.cfg, &h));
    load_all(h, n, 13);
    OK(nr_net_plan(h, 1, 16, 8, 8, 0));
    OK(nr_leaf_forward(h, nullptr, sample, nullptr, 0, out));
    int heads = 0, cores = 0;
    for (int i = 0; i < nr_net_num_ops(h); ++i) {
      heads += strstr(nr_net_op_desc(h, i), "tattn_head M=1024 C=640") != nullptr;
      cores += strstr(nr_net_op_desc(h, i), "attention mode=2") != nullptr;
    }
    CHECK(heads == 2 && cores == 0, "C = 640, F = 16 must plan one tattn_head launch per attention block");
    const long long bytes_first = nr_net_weight_bytes(h);
    const long live_before = nr_stub_live_allocs();
    for (auto& t : n.tensors)
      if (t.first.find("attention_blocks.0.to_q.weight") != std::string::npos) {
        int64_t numel = 1;
        for (auto d : t.second) numel *= d;
        std::vector<float> data((size_t)numel, 0.02f);
        OK(nr_net_load_tensor(h, t.first.c_str(), data.data(), t.second.data(), (int32_t)t.second.size()));
      }
    CHECK(nr_net_weight_bytes(h) < bytes_first, "a reloaded to_q must drop the stream / table / fold vectors built from it");
    CHECK(nr_leaf_forward(h, nullptr, sample, nullptr, 0, out) == NR_ERR_STATE, "forward after a reload must ask for a new plan");
    OK(nr_net_plan(h, 1, 16, 8, 8, 0));
    OK(nr_leaf_forward(h, nullptr, sample, nullptr, 0, out));
    CHECK(nr_net_weight_bytes(h) == bytes_first, "re-plan must rebuild exactly the dropped conversions");
    CHECK(nr_stub_live_allocs() == live_before, "reload + re-plan must not leak converted weights");
    OK(nr_net_plan(h, 1, 8, 8, 8, 0));                                // 8 frames: q|k|v GEMM + attention core + to_out (needs the folded matrix again)
    OK(nr_leaf_forward(h, nullptr, sample, nullptr, 0, out));
    cores = 0;
    for (int i = 0; i < nr_net_num_ops(h); ++i) cores += strstr(nr_net_op_desc(h, i), "attention mode=2") != nullptr;
    CHECK(cores == 2, "F = 8 is not eligible for the head kernel");
    OK(nr_net_plan(h, 1, 16, 8, 8, 0));                               // and back: stream and table are still cached
    OK(nr_leaf_forward(h, nullptr, sample, nullptr, 0, out));
    nr_net_destroy(h);
  }
  // ---------------- the other kinds: sgm U-Net, VAE decoder / encoder, CLIP ----------------
  if (nets.count("tiny_sgm")) {
    const Net& n = nets.at("tiny_sgm");
    nr_net* h = nullptr;
    OK(nr_net_create(&n.cfg, &h));
    load_all(h, n, 6);
    OK(nr_net_plan(h, 2, 1, 16, 16, 7));
    OK(nr_net_set_graph(h, 1));
    for (int i = 0; i < 70; ++i) OK(nr_sgm_unet_forward(h, nullptr, sample, 1.0f / (1.0f + i), ts, ctx, 7, cond, out));   // one graph per c_in: LRU churn
    nr_net_destroy(h);
  }
  for (const char* nm : {"tiny_vae_dec", "tiny_vae_enc"}) {
    if (!nets.count(nm)) continue;
    const Net& n = nets.at(nm);
    nr_net* h = nullptr;
    OK(nr_net_create(&n.cfg, &h));
    load_all(h, n, 7);
    const bool dec = std::string(nm) == "tiny_vae_dec";
    OK(nr_net_plan(h, 2, 1, dec ? 8 : 64, dec ? 8 : 64, 0));
    if (dec) OK(nr_vae_decode(h, nullptr, sample, 5.4f, 0.5f, 0.5f, 1, out));
    else OK(nr_vae_encode(h, nullptr, sample, 2.f, -1.f, out));
    nr_net_destroy(h);
  }
  if (nets.count("tiny_clip")) {
    const Net& n = nets.at("tiny_clip");
    nr_net* h = nullptr;
    OK(nr_net_create(&n.cfg, &h));
    load_all(h, n, 8);
    OK(nr_net_plan(h, 2, 1, 1, 77, 0));
    OK(nr_clip_text_forward(h, nullptr, reinterpret_cast<const int32_t*>(sample), out));
    CHECK(nr_net_plan(h, 2, 1, 1, 78, 0) == NR_ERR_ARG, "sequence beyond max_position_embeddings");
    nr_net_destroy(h);
  }
  CHECK(nr_stub_live_graphs() == 0, "graph executables leaked");
  printf("planner dry-run OK: %ld kernel launches enqueued, %ld graph captures, %ld device allocations still live (process-lifetime caches)\n",
         nr_stub_launches(), nr_stub_captures(), nr_stub_live_allocs());
  return 0;
}
