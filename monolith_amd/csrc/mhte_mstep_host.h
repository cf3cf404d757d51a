// Host side of the multi-table step and of the one-launch fused ops (kernels:
// mhte_mstep_kernels.h).  Included by mhte.hip, inside namespace-level scope, after Table and
// mhte_multi_table are defined.
//
// Reference call sites this replaces as ONE pair of launches per training step:
//   NT/multi_hash_table_ops.py:349-413 (lookup / apply_gradients over every table of the model),
//   RT/ops/multi_hash_table_lookup_op.cc:33-89, RT/ops/multi_hash_table_update_op.cc:47-100.
#ifndef MHTE_MSTEP_HOST_H_
#define MHTE_MSTEP_HOST_H_

namespace mhte {

// ---- device copies of the tables' views ---------------------------------------------------------
// (a view changes when the table doubles, gets a filter, ...: Table::view_version)
static void sync_views(mhte_multi_table* mt, hipStream_t st) {
  const size_t T = mt->tables.size();
  if (mt->view_uploaded.size() != T) {
    mt->d_views.reserve(T);
    mt->view_uploaded.assign(T, 0);
  }
  bool synced = false;
  for (size_t i = 0; i < T; ++i) {
    Table& tb = *mt->tables[i];
    if (mt->view_uploaded[i] == tb.view_version) continue;
    if (!synced) {  // launches in flight still read the old descriptors
      HIP_OK(hipStreamSynchronize(st));
      synced = true;
    }
    TableView v = tb.view;
    v.trace = nullptr;
    HIP_OK(hipMemcpy(mt->d_views.p + i, &v, sizeof(TableView), hipMemcpyHostToDevice));
    mt->view_uploaded[i] = tb.view_version;
  }
}

static uint32_t group_lanes(uint32_t dim) {
  uint32_t g = 8;
  while (g * 4 < dim && g < 64) g <<= 1;
  return g;
}

// Lane-group shape of a table in a launch (MHTE_SWITCH_GV): lanes per id | 1 when a lane moves one
// float.  float4 lanes need rows of whole float4s (Table::vec_ok) AND the table's slices of the flat
// buffers of the launch (float offsets off_a, off_b) on 16-byte boundaries; otherwise one float per
// lane, which covers rows of up to 64 floats — the reference's standard layouts with a dim-1 bias
// slice in front of the vector (NT/feature.py:117-120; distributed_ps_test.py:480-505: dims 17 / 33).
static uint32_t shape_code(const Table& tb, uint64_t off_a = 0, uint64_t off_b = 0) {
  const bool v4 = tb.vec_ok && off_a % 4 == 0 && off_b % 4 == 0;
  const Shape sh = pick_shape(tb.dim, v4);
  if (tb.dim > uint32_t(sh.G * sh.VEC))
    throw Error(MHTE_INVALID_ARGUMENT,
                "table " + tb.name + ": a row of " + std::to_string(tb.dim) + " floats that is not whole "
                "float4s on a 16-byte boundary of the flat buffer (a table of odd dim in front of it?) "
                "fits the fused step up to 64 floats");
  return uint32_t(sh.G) | (sh.VEC == 1 ? 1u : 0u);
}
static uint32_t shape_lanes(uint32_t code) { return code & ~3u; }
// ... of the segment kernels (fused optimize, the sharded step's owner side): bit 1 = the table has
// a whole-segment optimizer (GroupAdaGrad) — its own kernel instance (kShapeGroupBit)
static uint32_t seg_shape_code(const Table& tb, uint64_t off_a = 0) {
  return shape_code(tb, off_a) | (tb.has_group_opt ? uint32_t(kShapeGroupBit) : 0u);
}

// bump allocator over one hipMalloc (first pass with base == nullptr sizes it)
struct Arena {
  char* base = nullptr;
  size_t off = 0;
  template <class T>
  T* take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};

struct MultiStep {
  mhte_multi_table* mt = nullptr;
  int device = 0;
  uint32_t T = 0;
  int64_t max_batch = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  std::vector<MStepStatic> h_st;
  MStepStatic* d_st = nullptr;
  std::vector<uint64_t> st_version;     // Table::view_version the static descriptor was built for
  int cur = 0;                          // slot of the batch being trained
  int stage[2] = {0, 0};                // 0 empty, 1 deduplicated (scratch dirty), 2 numbered
  std::vector<uint32_t> n_slot[2];      // per table: batch size held by the slot
  int num_cus = 256;
  uint32_t ovs = 4;                     // workgroups launched per resident slot (T > 1)
  uint32_t scatter_ovs = 2;             // lookup workgroups launched per resident slot
  uint32_t item_target = 4 * kItemTarget;  // expected occurrences per heavy work item: bigger than
                                        // the single-table step's, whose items are its longest
                                        // chain; here the fixed round trips per item are what costs
  std::vector<uint64_t> fwd_epoch[2];   // Table::mut_epoch when the slot's batch was looked up
  bool has_hints[2] = {false, false};   // the slot's urow / uloc were written by a forward launch

  ~MultiStep() {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    if (arena) (void)hipFree(arena);
    if (d_st) (void)hipFree(d_st);
    if (side) (void)hipStreamDestroy(side);
    if (ev_now) (void)hipEventDestroy(ev_now);
    for (int i = 0; i < 2; ++i)
      if (ev_dedup[i]) (void)hipEventDestroy(ev_dedup[i]);
  }

  static uint32_t scratch_cap(int64_t n) {
    return 1u << std::max<uint32_t>(10, ceil_log2(uint64_t(2) * uint64_t(n)));
  }

  void layout(Arena& A, uint32_t t, MStepStatic& s) {
    const Table& tb = *mt->tables[t];
    const int64_t n = max_batch;
    const uint32_t C = scratch_cap(n);
    const uint32_t nblk = uint32_t((n + kRdBlock - 1) / kRdBlock);
    const uint32_t items = DedupWs::max_items(n);
    for (int sl = 0; sl < 2; ++sl) {
      RunView d{};
      d.hs = A.take<RdSlot>(size_t(C) + 2);
      d.hlist = A.take<uint32_t>((size_t(C) + 2) * kLightMax);
      d.cap_mask = C - 1;
      d.uslot = A.take<uint32_t>(size_t(n) + 1);
      d.ucnt = A.take<uint32_t>(size_t(n) + 1);
      d.upos = A.take<uint32_t>(size_t(n) + 1);
      d.btab_key = A.take<int64_t>(size_t(nblk) * kRdStride);
      d.btab_val = A.take<uint32_t>(size_t(nblk) * kRdStride);
      d.seg = A.take<uint16_t>(size_t(nblk) * kRdBlock);
      d.item_hdr = A.take<ItemHdr>(items);
      d.item_runs = A.take<uint32_t>(size_t(items) * 64);
      d.ctr = A.take<uint32_t>(4);
      d.ids = nullptr;
      d.n = 0;
      d.nblk = 0;
      d.item_target = item_target;
      d.uids = A.take<int64_t>(size_t(n) + 1);
      d.n_unique = A.take<uint32_t>(4);
      s.rv[sl] = d;
      s.part[sl] = A.take<float>(size_t(items) * tb.dim + 16);
      s.arrive[sl] = A.take<uint32_t>(size_t(n) + 2);
      s.urow[sl] = A.take<uint32_t>(size_t(n) + 2);
      s.uloc[sl] = A.take<unsigned long long>(size_t(n) + 2);
      s.uts[sl] = A.take<uint32_t>(size_t(n) + 2);
    }
    s.grad_u = A.take<float>(size_t(n) * tb.dim + 16);
    s.pending = A.take<uint32_t>(size_t(n) + 2);
    s.n_max = n;
    s.g = group_lanes(tb.dim);
    s.oneseg = tb.nseg == 1 ? 1u : 0u;
    s.nblk_build = DedupWs::build_blocks(s.rv[0]);
    s.count_hits = tb.count_hits ? 1u : 0u;
  }

  // sender_roles_only: the id-sharded step uses this object for dedup / numbering / scatter / sums; the
  // optimizers run on the owner (mhte_shard_host.h), so a whole-segment optimizer is no obstacle
  void init(mhte_multi_table* m, int64_t mb, bool sender_roles_only = false) {
    mt = m;
    device = m->device;
    T = uint32_t(m->tables.size());
    max_batch = mb;
    if (mb <= 0 || mb > int64_t(kRdMaxBlocks) * kRdBlock)
      throw Error(MHTE_INVALID_ARGUMENT, "multi step: max batch per table must be 1.." +
                                             std::to_string(kRdMaxBlocks * kRdBlock));
    for (uint32_t t = 0; t < T; ++t) {
      const Table& tb = *m->tables[t];
      if (!tb.fusable() && !(sender_roles_only && tb.fusable_shape()))
        throw Error(MHTE_INVALID_ARGUMENT,
                    "multi step: table " + tb.name + " does not fit the fused step (per-element "
                    "optimizers — GroupAdaGrad needs the whole segment; rows of whole float4s up to "
                    "256 floats, any other row layout up to 64)");
    }
    // A table moved with one float4 per lane needs its slice of the flat embedding / gradient buffers on
    // a 16-byte boundary, and the slice starts at the sum of n_t x dim_t of the tables in front of it —
    // a multiple of four floats for EVERY batch only when every earlier dim is one.  A row of more than
    // 64 floats has no one-float-per-lane form to fall back to (shape_code): such a model is refused
    // here, not by a throw in the middle of a step whose outcome depends on the batch sizes (in the
    // id-sharded step: after the exchanges were enqueued, with the peers left waiting).
    {
      bool odd_before = false;
      for (uint32_t t = 0; t < T; ++t) {
        const Table& tb = *m->tables[t];
        if (tb.dim > 64 && odd_before)
          throw Error(MHTE_INVALID_ARGUMENT,
                      "multi step: table " + tb.name + " (" + std::to_string(tb.dim) + " floats per row) follows, in "
                      "sorted-name order, a table whose dim is not a multiple of 4: its slice of the flat buffers "
                      "would leave its 16-byte alignment for some batch sizes, and a row of more than 64 floats "
                      "cannot fall back to one float per lane.  Rename the tables so that the wide ones come "
                      "first, or pad the odd dims");
        if (tb.dim % 4u) odd_before = true;
      }
    }
    {
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
        num_cus = cus;
    }
    if (const char* e = getenv("MHTE_MSTEP_OVERSUB")) ovs = std::max(1, atoi(e));
    if (const char* e = getenv("MHTE_MSTEP_SCATTER_OVS")) scatter_ovs = std::max(1, atoi(e));
    // (with a few tables in a launch the heavy items are the launch's longest chain again: cut them
    // as the single-table step does)
    item_target = T >= 4 ? 4 * kItemTarget : kItemTarget;
    if (const char* e = getenv("MHTE_MSTEP_ITEM_TARGET")) item_target = std::max<int>(int(kItemTarget), atoi(e));
    h_st.assign(T, MStepStatic{});
    Arena sizing;
    for (uint32_t t = 0; t < T; ++t) layout(sizing, t, h_st[t]);
    arena_bytes = sizing.off + 256;
    HIP_OK(hipMalloc(&arena, arena_bytes));
    HIP_OK(hipMemset(arena, 0, arena_bytes));
    Arena real;
    real.base = arena;
    for (uint32_t t = 0; t < T; ++t) layout(real, t, h_st[t]);
    HIP_OK(hipMalloc(&d_st, sizeof(MStepStatic) * T));
    HIP_OK(hipMemcpy(d_st, h_st.data(), sizeof(MStepStatic) * T, hipMemcpyHostToDevice));
    st_version.assign(T, 0);
    for (uint32_t t = 0; t < T; ++t) st_version[t] = m->tables[t]->view_version;
    n_slot[0].assign(T, 0);
    n_slot[1].assign(T, 0);
    fwd_epoch[0].assign(T, 0);
    fwd_epoch[1].assign(T, 0);
    clear_slots(3u, nullptr);
    make_streams();
    HIP_OK(hipDeviceSynchronize());
  }

  void clear_slots(uint32_t mask, hipStream_t st) {
    const uint32_t C = scratch_cap(max_batch);
    mstep_clear_kernel<<<dim3((C + 2 + 255) / 256, T), 256, 0, st>>>(ConstStatics(d_st), mask);
    HIP_OK(hipGetLastError());
  }

  // count_hits is the one per-table switch that can change after creation
  void sync_static(hipStream_t st) {
    for (uint32_t t = 0; t < T; ++t) {
      Table& tb = *mt->tables[t];
      if (st_version[t] == tb.view_version) continue;
      const uint32_t ch = tb.count_hits ? 1u : 0u;
      if (h_st[t].count_hits != ch) {
        h_st[t].count_hits = ch;
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpy(d_st + t, &h_st[t], sizeof(MStepStatic), hipMemcpyHostToDevice));
      }
      st_version[t] = tb.view_version;
    }
  }

  void check_ragged(const int64_t* split, int64_t n_split, const char* what) const {
    if (!split || n_split != int64_t(T) + 1)
      throw Error(MHTE_INVALID_ARGUMENT,
                  std::string("The length of tensor `") + what + "` doesn't equal to table num. " +
                      std::to_string(n_split - 1) + "v.s." + std::to_string(T));
    for (uint32_t t = 0; t < T; ++t) {
      const int64_t n = split[t + 1] - split[t];
      if (n < 0) throw Error(MHTE_INVALID_ARGUMENT, "id_split not monotonic");
      if (n > max_batch)
        throw Error(MHTE_INVALID_ARGUMENT, "multi step: table " + mt->tables[t]->name + " has " +
                                               std::to_string(n) + " ids, the step was created for " +
                                               std::to_string(max_batch));
    }
    if (split[T] - split[0] > int64_t(0xffffffffu))
      throw Error(MHTE_INVALID_ARGUMENT, "multi step: more than 2^32 ids");
  }

  static constexpr int kFwdBlock = 256;   // threads per workgroup of the lookup launch

  // lookup of the batch held (numbered) by `slot` into `out`: one launch per chunk of
  // kMaxStepTables tables
  void launch_fwd(float* out, int slot, hipStream_t st) {
    int64_t emb_off = 0;
    uint32_t active = 0;
    for (uint32_t t = 0; t < T; ++t) active += n_slot[slot][t] ? 1u : 0u;
    for (uint32_t t0 = 0; t0 < T; t0 += kMaxStepTables) {
      const uint32_t tc = std::min<uint32_t>(kMaxStepTables, T - t0);
      MFwdArgs A{};
      A.views = ConstViews(mt->d_views.p + t0);
      A.st = ConstStatics(d_st + t0);
      A.out = out;
      A.cur = uint32_t(slot);
      A.item_split = std::max<uint32_t>(1, item_target / kItemTarget);
      uint32_t gx = 0;
      for (uint32_t k = 0; k < tc; ++k) {
        const uint32_t t = t0 + k;
        MFwdTab& ft = A.tab[k];
        const Table& tb = *mt->tables[t];
        ft.n = n_slot[slot][t];
        if (uint64_t(emb_off) > 0xffffffffull)
          throw Error(MHTE_INVALID_ARGUMENT, "multi step: embedding buffer exceeds 2^32 floats");
        ft.emb_off = uint32_t(emb_off);
        emb_off += int64_t(ft.n) * tb.dim;
        ft.gv = shape_code(tb, uint64_t(ft.emb_off));
        if (ft.n) {
          // persistent workgroups: the launch's share of the resident slots (8 per CU) times the
          // oversubscription, split evenly over the tables; enough for one trip at most
          const uint32_t groups_per_wg = uint32_t(kFwdBlock) / shape_lanes(ft.gv);
          const uint32_t one_trip = (ft.n + groups_per_wg - 1) / groups_per_wg;
          const uint32_t share = std::max<uint32_t>(8, uint32_t(8 * num_cus) * scatter_ovs / std::max(1u, active));
          ft.nblk_s = std::max<uint32_t>(1, std::min(one_trip, share));
          gx = std::max(gx, ft.nblk_s);
        }
      }
      if (gx == 0) continue;
      const dim3 grid(gx, tc);
      A.trace = trace_region(kTagMStepFwd, grid.x * grid.y, kFwdBlock);
      // (one instance of the kernel per lane width among the tables: MHTE_SWITCH_G)
      bool w4 = false, w1 = false;
      for (uint32_t k = 0; k < tc; ++k)
        if (A.tab[k].n) ((A.tab[k].gv & 1u) ? w1 : w4) = true;
      if (w4) LAUNCH_HOT(kTagMStepFwd, (mstep_fwd_kernel<kFwdBlock, 4>), grid, kFwdBlock, st, A);
      if (w1) LAUNCH_HOT(kTagMStepFwd, (mstep_fwd_kernel<kFwdBlock, 1>), grid, kFwdBlock, st, A);
      HIP_OK(hipGetLastError());
    }
  }

  // lookup of the batch in `slot` + run dedup of (ids_next, split_next) into `slot_next`, ONE launch
  // (T <= kMaxStepTables; mstep_fwd_dedup_kernel)
  uint32_t fused_dedup_wgs = 0;   // persistent dedup workgroups of the fused launch (0: per CU count)
  uint32_t fused_lookup_wgs = 0;  // lookup workgroups of the fused launch over all tables (0: resident share)
  void launch_fwd_dedup(float* out, int slot, const int64_t* ids_next, const int64_t* split_next,
                        int slot_next, hipStream_t st) {
    MFwdArgs A{};
    A.views = ConstViews(mt->d_views.p);
    A.st = ConstStatics(d_st);
    A.out = out;
    A.cur = uint32_t(slot);
    A.item_split = std::max<uint32_t>(1, item_target / kItemTarget);
    MDedupArgs D{};
    D.st = ConstStatics(d_st);
    D.ids = ids_next + split_next[0];
    D.slot = uint32_t(slot_next);
    D.T = T;
    MFwdFuse F{};
    uint32_t active = 0, dblocks = 0;
    for (uint32_t t = 0; t < T; ++t) active += n_slot[slot][t] ? 1u : 0u;
    int64_t emb_off = 0;
    uint32_t lin = 0, gx1 = 0;
    // lookup workgroups of 1024 threads: two resident per CU; the dedup's persistent ones take
    // their share of the slots, the rest goes to the tables in proportion
    const uint32_t slots = uint32_t(2 * num_cus);
    for (uint32_t t = 0; t < T; ++t) {
      D.id_off[t] = uint32_t(split_next[t] - split_next[0]);
      D.blk_start[t] = dblocks;
      dblocks += uint32_t((split_next[t + 1] - split_next[t] + kRdBlock - 1) / kRdBlock);
    }
    D.id_off[T] = uint32_t(split_next[T] - split_next[0]);
    D.blk_start[T] = dblocks;
    F.nd = std::min<uint32_t>(dblocks, fused_dedup_wgs ? fused_dedup_wgs : uint32_t(num_cus) / 2);
    uint32_t room = std::max<uint32_t>(slots > F.nd ? slots - F.nd : 8u, 8u) * scatter_ovs;
    if (fused_lookup_wgs) room = fused_lookup_wgs;
    for (uint32_t t = 0; t < T; ++t) {
      MFwdTab& ft = A.tab[t];
      const Table& tb = *mt->tables[t];
      ft.n = n_slot[slot][t];
      if (uint64_t(emb_off) > 0xffffffffull)
        throw Error(MHTE_INVALID_ARGUMENT, "multi step: embedding buffer exceeds 2^32 floats");
      ft.emb_off = uint32_t(emb_off);
      emb_off += int64_t(ft.n) * tb.dim;
      F.fwd_start[t] = lin;
      ft.gv = shape_code(tb, uint64_t(ft.emb_off));
      if (ft.n && (ft.gv & 1u)) {
        // one float per lane: looked up by mstep_fwd_kernel<.., 1> below (no blocks in the fused launch)
        const uint32_t groups_per_wg = uint32_t(kFwdBlock) / shape_lanes(ft.gv);
        const uint32_t one_trip = (ft.n + groups_per_wg - 1) / groups_per_wg;
        const uint32_t share = std::max<uint32_t>(8, uint32_t(8 * num_cus) * scatter_ovs / std::max(1u, active));
        ft.nblk_s = std::max<uint32_t>(1, std::min(one_trip, share));
        gx1 = std::max(gx1, ft.nblk_s);
      } else if (ft.n) {
        const uint32_t groups_per_wg = uint32_t(kRdBlock) / shape_lanes(ft.gv);
        const uint32_t one_trip = (ft.n + groups_per_wg - 1) / groups_per_wg;
        const uint32_t share = std::max<uint32_t>(2, room / std::max(1u, active));
        ft.nblk_s = std::max<uint32_t>(1, std::min(one_trip, share));
        lin += ft.nblk_s;
      }
    }
    F.fwd_start[T] = lin;
    const uint32_t grid = F.nd + lin;
    if (gx1) {
      MFwdArgs A1 = A;
      A1.trace = trace_region(kTagMStepFwd, gx1 * T, kFwdBlock);
      LAUNCH_HOT(kTagMStepFwd, (mstep_fwd_kernel<kFwdBlock, 1>), dim3(gx1, T), kFwdBlock, st, A1);
      HIP_OK(hipGetLastError());
    }
    if (!grid) return;
    // (the dedup workgroups FIRST: spread evenly among the lookups' they measured 162-502 us against
    // 161 us — MHTE_MSTEP_FUSE_INTERLEAVE=1 keeps the other mapping for A/B runs)
    static const bool interleave = getenv("MHTE_MSTEP_FUSE_INTERLEAVE") != nullptr;
    F.period = (interleave && F.nd) ? std::max<uint32_t>(1, grid / F.nd) : 1u;
    A.trace = trace_region(kTagMStepFwd, grid, kRdBlock);
    D.trace = A.trace;
    LAUNCH_HOT(kTagMStepFwd, mstep_fwd_dedup_kernel, grid, kRdBlock, st, A, D, F);
    HIP_OK(hipGetLastError());
  }

  // run dedup of the ragged batch (ids, split) into `slot`; max_wgs: persistent workgroups per
  // launch (0: one per item)
  void launch_dedup(const int64_t* ids, const int64_t* split, int slot, uint32_t max_wgs, hipStream_t st) {
    for (uint32_t t0 = 0; t0 < T; t0 += kMaxStepTables) {
      const uint32_t tc = std::min<uint32_t>(kMaxStepTables, T - t0);
      MDedupArgs A{};
      A.st = ConstStatics(d_st + t0);
      A.ids = ids + split[t0];
      A.slot = uint32_t(slot);
      A.T = tc;
      uint32_t blocks = 0;
      for (uint32_t k = 0; k < tc; ++k) {
        const uint32_t t = t0 + k;
        A.id_off[k] = uint32_t(split[t] - split[t0]);
        A.blk_start[k] = blocks;
        blocks += uint32_t((split[t + 1] - split[t] + kRdBlock - 1) / kRdBlock);
      }
      A.id_off[tc] = uint32_t(split[t0 + tc] - split[t0]);
      A.blk_start[tc] = blocks;
      if (!blocks) continue;
      const uint32_t grid = max_wgs ? std::min(blocks, max_wgs) : blocks;
      A.trace = trace_region(kTagDedup, grid, kRdBlock);
      LAUNCH_HOT(kTagDedup, mstep_dedup_kernel, grid, kRdBlock, st, A);
      HIP_OK(hipGetLastError());
    }
  }

  struct BwdPlan {
    int64_t global_step = 0;
    const float* grads = nullptr;
    const int64_t* split = nullptr;   // of the batch in slot `cur` (nullptr: build only)
    const float* lrs = nullptr;
    int64_t update_time = 0;
    bool exact_order = false;
  };

  // one backward launch per chunk: apply of slot `slot_cur` (when p.grads) | numbering of slot
  // slot_cur ^ 1 (when build_next); then the displacement pass
  void launch_bwd(const BwdPlan& p, int slot_cur, bool build_next, hipStream_t st) {
    int64_t grad_off = 0;
    int64_t lr_off = 0;
    uint32_t active = 0;
    if (p.grads)
      for (uint32_t t = 0; t < T; ++t) active += n_slot[slot_cur][t] ? 1u : 0u;
    const uint32_t budget = uint32_t(kBwdBlocksPerCu * num_cus);
    const uint32_t share = std::max<uint32_t>(32, budget * (active > 1 ? ovs : 1u) / std::max(1u, active));
    for (uint32_t t0 = 0; t0 < T; t0 += kMaxStepTables) {
      const uint32_t tc = std::min<uint32_t>(kMaxStepTables, T - t0);
      MBwdArgs A{};
      A.views = ConstViews(mt->d_views.p + t0);
      A.st = ConstStatics(d_st + t0);
      A.grads = p.grads;
      A.cur = uint32_t(slot_cur);
      uint32_t gx = 0;
      bool any_apply = false, any_exact = false;
      for (uint32_t k = 0; k < tc; ++k) {
        const uint32_t t = t0 + k;
        MBwdTab& bt = A.tab[k];
        const Table& tb = *mt->tables[t];
        const uint32_t n = p.grads ? n_slot[slot_cur][t] : 0u;
        bt.build_next = (build_next && n_slot[slot_cur ^ 1][t]) ? 1u : 0u;
        bt.full = tb.basic_opts() ? 0u : 1u;
        bt.n = n;
        bt.n_next = n_slot[slot_cur ^ 1][t];
        uint32_t blocks = bt.build_next ? h_st[t].nblk_build : 0u;
        if (n) {
          bt.apply = 1;
          any_apply = true;
          bt.grad_off = uint32_t(grad_off);
          grad_off += int64_t(n) * tb.dim;
          for (int i = 0; i < kMaxSegments; ++i)
            bt.a.lr[i] = (i < int(tb.nseg)) ? p.lrs[lr_off + i] : 0.f;
          bt.a.ts = static_cast<uint32_t>(p.update_time);
          bt.a.sum_dups = 1;
          bt.a.filter_mode = 1;
          bt.a.global_step = p.global_step;
          // MHTE_EXACT_ORDER: the heavy lists' sums come from mstep_exact_sum_kernel, launched in front (as in the
          // single-table step; MHTE_EXACT_WALK=1: round 5's form, every list walked by its lane group)
          static const bool exact_walk = getenv("MHTE_EXACT_WALK") != nullptr && atoi(getenv("MHTE_EXACT_WALK")) != 0;
          const bool exact_pre = p.exact_order && !exact_walk && tb.dim <= 256u;
          const bool exact_old = p.exact_order && !exact_pre;
          if (exact_pre) {
            bt.apply |= 2u;
            any_exact = true;
          }
          bt.light_max = exact_old ? 0xffffffffu : uint32_t(kStepLightMax);
          bt.hints = (has_hints[slot_cur] && fwd_epoch[slot_cur][t] == tb.mut_epoch) ? 1u : 0u;
          bt.gv = shape_code(tb, uint64_t(bt.grad_off));
          const uint32_t groups_per_wg = 256u / shape_lanes(bt.gv);
          const uint32_t cap_items = DedupWs::max_items(n);
          bt.nblk_items = exact_old ? 0u
                                    : std::min<uint32_t>(cap_items, std::min<uint32_t>(
                                              uint32_t(num_cus) * 10 / 8, std::max<uint32_t>(8, share / 4)));
          const uint32_t need = (n + groups_per_wg - 1) / groups_per_wg;
          const uint32_t room = share > bt.nblk_items + std::min<uint32_t>(128, share / 8) + 16
                                    ? share - bt.nblk_items - std::min<uint32_t>(128, share / 8)
                                    : 16u;
          bt.nblk_ids = std::max<uint32_t>(1, std::min(need, room));
          blocks += bt.nblk_items + bt.nblk_ids;
        }
        lr_off += tb.nseg;
        gx = std::max(gx, blocks);
      }
      if (gx == 0) continue;
      if (any_exact) {   // (one 135-KB workgroup per CU: the chip's CUs dealt over the launch's tables)
        mstep_exact_sum_kernel<<<dim3(std::max<uint32_t>(4, uint32_t(num_cus) / tc), tc), kExactThreads, 0, st>>>(A);
        HIP_OK(hipGetLastError());
      }
      A.trace = trace_region(kTagMStepBwd, gx * tc, 256);
      // one launch per optimizer family present (the BASIC instantiation keeps the register budget
      // of SGD / Adagrad / FTRL tables; a workgroup of the other family's table leaves at once)
      // ... and per lane width (MHTE_SWITCH_G; a table with nothing to apply has shape code 0 = the
      // float4 instance, which then runs its numbering)
      bool fam[2][2][2] = {};
      for (uint32_t k = 0; k < tc; ++k)
        fam[A.tab[k].full ? 1 : 0][A.tab[k].gv & 1u][h_st[t0 + k].oneseg ? 1 : 0] = true;
      // ... and the instances with the admission filter's code only when a table of the launch has a filter
      bool filt = false;
      for (uint32_t k = 0; k < tc; ++k) filt = filt || mt->tables[t0 + k]->flt_slots != nullptr;
#define MHTE_BWD_LAUNCH(F_, W_, O_)                                                                              \
  if (fam[F_][W_ == 1][O_]) {                                                                                    \
    if (filt) LAUNCH_HOT(kTagMStepBwd, (mstep_bwd_kernel<F_ != 0, W_, O_ != 0, true>), dim3(gx, tc), 256, st, A); \
    else LAUNCH_HOT(kTagMStepBwd, (mstep_bwd_kernel<F_ != 0, W_, O_ != 0, false>), dim3(gx, tc), 256, st, A);   \
  }
      MHTE_BWD_LAUNCH(0, 4, 1);
      MHTE_BWD_LAUNCH(0, 4, 0);
      MHTE_BWD_LAUNCH(0, 1, 1);
      MHTE_BWD_LAUNCH(0, 1, 0);
      MHTE_BWD_LAUNCH(1, 4, 1);
      MHTE_BWD_LAUNCH(1, 4, 0);
      MHTE_BWD_LAUNCH(1, 1, 1);
      MHTE_BWD_LAUNCH(1, 1, 0);
#undef MHTE_BWD_LAUNCH
      HIP_OK(hipGetLastError());
      if (any_apply) {
        mstep_slow_kernel<<<tc, 64 * kSlowWaves, 0, st>>>(A);
        HIP_OK(hipGetLastError());
        for (uint32_t k = 0; k < tc; ++k)
          if (A.tab[k].apply) ++mt->tables[t0 + k]->mut_epoch;
        uint64_t asked = 0;   // ids the launch can have put to the filter
        for (uint32_t k = 0; k < tc; ++k)
          if (A.tab[k].apply && mt->tables[t0 + k]->flt_slots) asked += A.tab[k].n;
        for (uint32_t k = 0; k < tc; ++k)
          if (A.tab[k].apply && mt->tables[t0 + k]->flt_slots) {
            mt->tables[t0 + k]->filter_maintain(st, asked);   // (one filter for all tables: once is enough)
            break;
          }
      }
    }
  }

  // dedup + numbering of (ids, split) into `slot`, on its own (first batch of a pipeline)
  void dedup_now(const int64_t* ids, const int64_t* split, int slot, hipStream_t st) {
    if (stage[slot] == 1) clear_slots(1u << slot, st);
    for (uint32_t t = 0; t < T; ++t) n_slot[slot][t] = uint32_t(split[t + 1] - split[t]);
    has_hints[slot] = false;
    launch_dedup(ids, split, slot, 0, st);
    BwdPlan none;
    launch_bwd(none, slot ^ 1, true, st);
    stage[slot] = 2;
  }

  // ---- streams: the dedup of the next batch runs on a stream of its own beside the step's launches
  hipStream_t side = nullptr;
  hipEvent_t ev_now = nullptr;            // main stream, at the start of a forward call
  hipEvent_t ev_dedup[2] = {nullptr, nullptr};   // side: the slot's dedup has finished
  bool dedup_on_side[2] = {false, false};
  uint32_t dedup_wgs = 128;               // persistent workgroups of the side-stream dedup
  bool use_side = false;                  // MHTE_MSTEP_SIDE=1 (measured slower: the two queues
                                          // compete for the dispatcher)
  bool fuse_fwd = true;                   // lookup + next batch's dedup in one launch
                                          // (MHTE_MSTEP_FUSE_FWD=0: two launches)

  void make_streams() {
    HIP_OK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&ev_now, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIP_OK(hipEventCreateWithFlags(&ev_dedup[i], hipEventDisableTiming));
    if (const char* e = getenv("MHTE_MSTEP_DEDUP_WGS")) dedup_wgs = std::max(1, atoi(e));
    if (const char* e = getenv("MHTE_MSTEP_SIDE")) use_side = atoi(e) != 0;
    if (const char* e = getenv("MHTE_MSTEP_FUSE_FWD")) fuse_fwd = atoi(e) != 0;
    if (const char* e = getenv("MHTE_MSTEP_FUSE_DEDUP_WGS")) fused_dedup_wgs = uint32_t(std::max(1, atoi(e)));
    if (const char* e = getenv("MHTE_MSTEP_FUSE_LOOKUP_WGS")) fused_lookup_wgs = uint32_t(std::max(1, atoi(e)));
  }

  // main stream waits for the side-stream dedup of `slot` (if that is where it ran)
  void join_dedup(int slot, hipStream_t st) {
    if (dedup_on_side[slot]) {
      HIP_OK(hipStreamWaitEvent(st, ev_dedup[slot], 0));
      dedup_on_side[slot] = false;
    }
  }

  void forward(const int64_t* id, const int64_t* split, int64_t n_split, float* emb, int64_t emb_len,
               const int64_t* id_next, const int64_t* split_next, int64_t n_split_next,
               int prefetched, hipStream_t st) {
    check_ragged(split, n_split, "id");
    if (id_next) check_ragged(split_next, n_split_next, "id_next");
    if (!id || !emb) throw Error(MHTE_INVALID_ARGUMENT, "multi step forward: null argument");
    if (!aligned16(emb)) throw Error(MHTE_INVALID_ARGUMENT, "multi step: embedding must be 16-byte aligned");
    int64_t need = 0;
    for (uint32_t t = 0; t < T; ++t) need += (split[t + 1] - split[t]) * int64_t(mt->tables[t]->dim);
    if (need > emb_len)
      throw Error(MHTE_INVALID_ARGUMENT, "embedding buffer too short: need " + std::to_string(need));
    for (auto& tb : mt->tables) tb->finish_pending(st);
    sync_views(mt, st);
    sync_static(st);
    if (prefetched) {
      const int nxt = cur ^ 1;
      bool same = stage[nxt] >= 1;
      for (uint32_t t = 0; same && t < T; ++t) same = n_slot[nxt][t] == uint32_t(split[t + 1] - split[t]);
      if (!same)
        throw Error(MHTE_FAILED_PRECONDITION,
                    "multi step forward: this batch was not deduplicated ahead by the previous forward");
      cur = nxt;
    }
    const int nxt = cur ^ 1;
    auto dedup_next = [&](hipStream_t ds, uint32_t wgs) {
      if (stage[nxt] == 1) clear_slots(1u << nxt, ds);
      for (uint32_t t = 0; t < T; ++t) n_slot[nxt][t] = uint32_t(split_next[t + 1] - split_next[t]);
      has_hints[nxt] = false;
      launch_dedup(id_next, split_next, nxt, wgs, ds);
      stage[nxt] = 1;
    };
    if (id_next && use_side) {
      // The next batch's dedup on the side stream, beside this step's lookup and update.  It
      // starts after everything the caller has enqueued so far: the ids may have just been
      // produced, and the launches that last read or wrote the slot are among it.
      HIP_OK(hipEventRecord(ev_now, st));
      HIP_OK(hipStreamWaitEvent(side, ev_now, 0));
      dedup_next(side, dedup_wgs);
      HIP_OK(hipEventRecord(ev_dedup[nxt], side));
      dedup_on_side[nxt] = true;
    }
    if (prefetched) {
      if (stage[cur] == 1) {  // (forward, forward: the batch was never numbered)
        join_dedup(cur, st);
        BwdPlan none;
        launch_bwd(none, cur ^ 1, true, st);
        stage[cur] = 2;
      }
    } else {
      join_dedup(cur, st);   // (a dedup still running into this slot is overwritten in order)
      dedup_now(id, split, cur, st);
    }
    // the lookup reads the ids from the slot's numbering (distinct ids + occurrence runs), not
    // from `id`: the caller's promise (prefetched) is that they are the same batch
    if (id_next && !use_side && fuse_fwd && T <= uint32_t(kMaxStepTables)) {
      // lookup of this batch and the run dedup of the next one in ONE launch
      if (stage[nxt] == 1) clear_slots(1u << nxt, st);
      for (uint32_t t = 0; t < T; ++t) n_slot[nxt][t] = uint32_t(split_next[t + 1] - split_next[t]);
      has_hints[nxt] = false;
      launch_fwd_dedup(emb, cur, id_next, split_next, nxt, st);
      stage[nxt] = 1;
      for (uint32_t t = 0; t < T; ++t) fwd_epoch[cur][t] = mt->tables[t]->mut_epoch;
      has_hints[cur] = true;
      return;
    }
    launch_fwd(emb, cur, st);
    for (uint32_t t = 0; t < T; ++t) fwd_epoch[cur][t] = mt->tables[t]->mut_epoch;
    has_hints[cur] = true;
    if (id_next && !use_side) dedup_next(st, 0);  // in stream order, behind the lookup
  }

  void backward(const float* grads, int64_t grads_len, const float* lrs, int64_t n_lr,
                int64_t update_time, bool exact_order, hipStream_t st, int64_t global_step = 0) {
    if (stage[cur] == 0)
      throw Error(MHTE_FAILED_PRECONDITION, "multi step backward: no forward batch outstanding");
    if (!grads || !lrs) throw Error(MHTE_INVALID_ARGUMENT, "multi step backward: null argument");
    if (!aligned16(grads)) throw Error(MHTE_INVALID_ARGUMENT, "multi step: gradients must be 16-byte aligned");
    int64_t need = 0, need_lr = 0;
    for (uint32_t t = 0; t < T; ++t) {
      need += int64_t(n_slot[cur][t]) * int64_t(mt->tables[t]->dim);
      need_lr += mt->tables[t]->nseg;
    }
    if (need > grads_len)
      throw Error(MHTE_INVALID_ARGUMENT, "The length of tensor `value` is too short. Currently value" +
                                             std::to_string(grads_len));
    if (need_lr > n_lr)
      throw Error(MHTE_INVALID_ARGUMENT,
                  "The length of tensor `learning_rate` is too short. Currently value" + std::to_string(n_lr));
    for (uint32_t t = 0; t < T; ++t) {
      Table& tb = *mt->tables[t];
      tb.finish_pending(st);
      if (!n_slot[cur][t]) continue;
      tb.note_update_time(update_time);
      tb.ensure_capacity(uint64_t(n_slot[cur][t]), st);
    }
    sync_views(mt, st);
    sync_static(st);
    if (stage[cur] == 1) {  // (forward, forward, backward: the batch was never numbered)
      join_dedup(cur, st);
      BwdPlan none;
      launch_bwd(none, cur ^ 1, true, st);
      stage[cur] = 2;
    }
    BwdPlan p;
    p.grads = grads;
    p.lrs = lrs;
    p.update_time = update_time;
    p.exact_order = exact_order;
    p.global_step = global_step;
    const int nxt = cur ^ 1;
    const bool build_next = stage[nxt] == 1;
    if (build_next) join_dedup(nxt, st);
    launch_bwd(p, cur, build_next, st);
    for (uint32_t t = 0; t < T; ++t)
      if (n_slot[cur][t]) mt->tables[t]->maybe_evict(st);  // (cadence of the reference's bridge)
    stage[cur] = 0;
    has_hints[cur] = false;
    if (build_next) stage[nxt] = 2;
  }
};

// =================================================================================================
// One-launch fused ops
// =================================================================================================
// true when every table can take the segment kernels (rows of whole float4s, or up to 64 floats of
// any layout; tables with a whole-segment optimizer — GroupAdaGrad — go to their own instance)
static bool seg_shapes_ok(const mhte_multi_table* t) {
  for (auto& tb : t->tables) {
    Shape sh = pick_shape(tb->dim, tb->vec_ok);   // (other row layouts: one float per lane, up to 64)
    if (tb->dim > uint32_t(sh.G * sh.VEC)) return false;
  }
  return true;
}
// ... and the model fits one launch of the fused lookup / optimize ops (the id-sharded step chunks its
// launches instead: mhte_shard_host.h)
static bool seg_kernels_ok(const mhte_multi_table* t) {
  return t->tables.size() <= size_t(kMaxStepTables) && seg_shapes_ok(t);
}

static void fused_lookup_segments(mhte_multi_table* t, const int64_t* ids, const int32_t* ko,
                                  const int32_t* eo, int nseg_all, float* embeddings, hipStream_t st) {
  const int T = int(t->tables.size());
  for (auto& tb : t->tables) tb->finish_pending(st);
  sync_views(t, st);
  const int per = std::max(T, (kMaxSegs / T) * T);  // whole shards per launch
  for (int s0 = 0; s0 < nseg_all; s0 += per) {
    const int ns = std::min(per, nseg_all - s0);
    SegLookupArgs A{};
    A.views = ConstViews(t->d_views.p);
    A.ids = ids;
    A.out = embeddings;
    A.T = uint32_t(T);
    A.seg0 = uint32_t(s0);
    uint32_t gx = 0;
    for (int y = 0; y <= ns; ++y) {
      A.id_off[y] = uint32_t(ko[s0 + y]);
      A.emb_off[y] = uint32_t(eo[s0 + y]);
    }
    for (int k = 0; k < T; ++k) {
      // (float4 lanes only if every segment of the table starts on a 16-byte boundary of `embeddings`)
      uint64_t worst = 0;
      for (int y = 0; y < ns; ++y)
        if ((s0 + y) % T == k && (A.emb_off[y] % 4u)) worst = 1;
      A.g[k] = uint8_t(shape_code(*t->tables[k], worst));
      A.count_hits[k] = t->tables[k]->count_hits ? 1 : 0;
    }
    for (int y = 0; y < ns; ++y) {
      const uint32_t n = A.id_off[y + 1] - A.id_off[y];
      const uint32_t g = shape_lanes(A.g[(s0 + y) % T]);
      gx = std::max(gx, uint32_t((uint64_t((n + 1) / 2) * g + 511) / 512));
    }
    if (gx == 0) continue;
    bool w4 = false, w1 = false;
    for (int k = 0; k < T; ++k) ((A.g[k] & 1u) ? w1 : w4) = true;
    if (w4) LAUNCH_HOT(kTagLookup, seg_lookup_kernel<4>, dim3(gx, ns), 512, st, A);
    if (w1) LAUNCH_HOT(kTagLookup, seg_lookup_kernel<1>, dim3(gx, ns), 512, st, A);
    HIP_OK(hipGetLastError());
  }
}

// FusedOptimize over segments whose ids are pairwise distinct (MHTE_IDS_UNIQUE): one upsert launch
// per kMaxSegs segments + one displacement launch.
static void fused_optimize_segments(mhte_multi_table* t, const int64_t* ids,
                                    const int32_t* fused_slot_size, const float* id_grads,
                                    const int32_t* id_offsets, const int32_t* grad_offsets,
                                    const float* learning_rates, int64_t req_time,
                                    int64_t global_step, int num_of_shards, hipStream_t st) {
  const int T = int(t->tables.size());
  const int nseg_all = T * num_of_shards;
  std::vector<uint64_t> per_table(size_t(T), 0);
  for (int y = 0; y < nseg_all; ++y) per_table[size_t(y % T)] += uint64_t(fused_slot_size[y]);
  for (int k = 0; k < T; ++k) {
    Table& tb = *t->tables[k];
    tb.finish_pending(st);
    if (!per_table[size_t(k)]) continue;
    tb.note_update_time(req_time);
    tb.ensure_capacity(per_table[size_t(k)], st);
    tb.pending.reserve(2 * size_t(per_table[size_t(k)]) + 2);
  }
  sync_views(t, st);
  const int per = std::max(T, (kMaxSegs / T) * T);
  for (int s0 = 0; s0 < nseg_all; s0 += per) {
    const int ns = std::min(per, nseg_all - s0);
    SegUpsertArgs A{};
    A.views = ConstViews(t->d_views.p);
    A.ids = ids;
    A.grads = id_grads;
    A.T = uint32_t(T);
    A.seg0 = uint32_t(s0);
    A.nseg = uint32_t(ns);
    uint32_t gx = 0;
    int64_t lr_off = 0;
    for (int k = 0; k < T; ++k) {
      Table& tb = *t->tables[k];
      {
        uint64_t worst = 0;
        for (int y = 0; y < ns; ++y)
          if ((s0 + y) % T == k && (uint32_t(grad_offsets[s0 + y]) % 4u)) worst = 1;
        A.g[k] = uint8_t(seg_shape_code(tb, worst));
      }
      A.pending[k] = tb.pending.p;
      ApplyArgs& a = A.a[k];
      for (int i = 0; i < kMaxSegments; ++i)
        a.lr[i] = (i < int(tb.nseg)) ? learning_rates[lr_off + i] : 0.f;
      lr_off += tb.nseg;  // (restarts per shard in the reference: the same slice for every shard)
      a.ts = static_cast<uint32_t>(req_time);
      a.sum_dups = 0;
      a.filter_mode = 0;
      a.global_step = global_step;
    }
    for (int y = 0; y < ns; ++y) {
      const uint32_t n = uint32_t(fused_slot_size[s0 + y]);
      A.id_off[y] = uint32_t(id_offsets[s0 + y]);
      A.grad_off[y] = uint32_t(grad_offsets[s0 + y]);
      if (y == ns - 1) A.id_off[ns] = A.id_off[y] + n;
      else if (uint32_t(id_offsets[s0 + y + 1]) != A.id_off[y] + n)
        throw Error(MHTE_INVALID_ARGUMENT, "id_offsets do not follow fused_slot_size");
      const uint32_t gl = shape_lanes(A.g[(s0 + y) % T]);
      gx = std::max(gx, (n + 256 / gl - 1) / (256 / gl));
    }
    if (gx == 0) continue;
    gx = std::min<uint32_t>(gx, 1024);
    bool inst[2][2] = {};   // [one float per lane][whole-segment optimizer]
    for (int k = 0; k < T; ++k) inst[A.g[k] & 1u][(A.g[k] >> 1) & 1u] = true;
    if (inst[0][0]) LAUNCH_HOT(kTagUpsert, (seg_upsert_kernel<4, false>), dim3(gx, ns), 256, st, A);
    if (inst[1][0]) LAUNCH_HOT(kTagUpsert, (seg_upsert_kernel<1, false>), dim3(gx, ns), 256, st, A);
    if (inst[0][1]) LAUNCH_HOT(kTagUpsert, (seg_upsert_kernel<4, true>), dim3(gx, ns), 256, st, A);
    if (inst[1][1]) LAUNCH_HOT(kTagUpsert, (seg_upsert_kernel<1, true>), dim3(gx, ns), 256, st, A);
    seg_slow_kernel<<<T, 64, 0, st>>>(A);
    HIP_OK(hipGetLastError());
    for (int k = 0; k < T; ++k)
      if (per_table[size_t(k)]) ++t->tables[k]->mut_epoch;
  }
  for (int k = 0; k < T; ++k)
    if (per_table[size_t(k)]) t->tables[k]->maybe_evict(st);
}

}  // namespace mhte
#endif  // MHTE_MSTEP_HOST_H_
