// Host side of the id-sharded multi-table step (kernels + wire format: mhte_shard_kernels.h).
// Included by mhte.hip after mhte_mstep_host.h.
//
// One process per GPU; rank r owns {id : floormod(id, N) == r} of EVERY table of the model
// (NT/distributed_ps.py:289) as a complete local mhte_multi_table.  A step on a rank:
//
//   forward   owner lookup of the received id blocks: rows straight into     1 launch
//               the senders' row blocks, a record per id for the update
//               (+ the displacement pass the last update owes: one rank)
//             sync point (peer stores)  |  RCCL group of N send/recv pairs     1 launch | 1 group
//             scatter rows to the occurrences | run dedup of the NEXT batch    1 launch
//               (depends on ids only, NT/distributed_ps_sync.py:199-203)
//   backward  per-id gradient sums straight into the owners' gradient blocks | 1 launch
//               numbering + owner packing of the next batch
//             sync point  |  RCCL groups (gradients, next id blocks)           1 launch | 2 groups
//             owner: every sender's block, rank order per id, ONE launch       1 launch
//             displacement pass of everything that launch deferred             1 launch (N > 1, or FULL rows)
//   7 launches with peer stores at any N; 4-5 with one rank and no transport.  (Round 4: a push and a sync
//   launch per exchange and an upsert + displacement pair per sender: 11 at N = 2, 23 at N = 8.  Tables with
//   an admission filter and MHTE_SHARD_PER_PEER=1 keep the per-sender pair; MHTE_SHARD_DIRECT=0 the pushes.)
//   (a batch that was not prepared ahead costs two more launches in its forward)
//
// Everything the host decides is a function of the configured capacities: no count crosses to the
// host, so there is no D2H copy and no stream synchronisation in the step.  RCCL is called directly
// (ncclSend / ncclRecv in one group on the caller's stream), loaded at run time from the librccl
// the process already has.  With world == 1 and no communicator the exchange is the identity: send
// and receive buffers are the same memory.
#ifndef MHTE_SHARD_HOST_H_
#define MHTE_SHARD_HOST_H_

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace mhte {

// ---- RCCL, bound at run time ------------------------------------------------------------------------
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // (optional: diagnostics)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;

  static Rccl& get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
      std::vector<std::string> names;
      if (const char* e = getenv("MHTE_RCCL_LIBRARY")) names.push_back(e);
      for (const char* n : {"librccl.so.1", "librccl.so"}) names.push_back(n);
      // a copy the process already holds (PyTorch ships its own) first: one RCCL per process
      for (auto& n : names)
        if (!r.h) r.h = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD);
      for (auto& n : names)
        if (!r.h) r.h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
      if (!r.h) return;
      auto sym = [&](const char* s) { return dlsym(r.h, s); };
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
      r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
      r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
      r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
      r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
      r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
      r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
    });
    if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd ||
        !r.Send || !r.Recv)
      throw Error(MHTE_UNAVAILABLE, "RCCL (librccl.so) could not be loaded: set MHTE_RCCL_LIBRARY");
    return r;
  }
  void ok(ncclResult_t e, const char* what) const {
    if (e != ncclSuccess)
      throw Error(MHTE_INTERNAL, std::string("RCCL ") + what + ": " +
                                     (GetErrorString ? GetErrorString(e) : "error"));
  }
};

enum ShardExchange : int { kXIds = 0, kXRows = 1, kXGrads = 2 };

struct ShardFwd {
  const int64_t* id = nullptr;
  const int64_t* split = nullptr;
  float* emb = nullptr;
  int64_t emb_len = 0;
  const int64_t* id_next = nullptr;
  const int64_t* split_next = nullptr;
};

struct ShardStep {
  MultiStep ms;                 // sender side: run dedup + numbering of this rank's batches
  mhte_multi_table* mt = nullptr;
  int device = 0;
  int rank = 0, world = 1;
  uint32_t T = 0;
  ShardGeom geo{};
  std::vector<ShardTab> tab;
  ShardTab* d_tab = nullptr;    // the same on the device (push / cvt kernels)
  mutable std::vector<ShardGatherTab> gt_all;   // gather_tabs: all T tables of a launch group
  struct OwnerChunk {           // owner_apply: the arguments of tables [c * kMaxStepTables, ...)
    ShardOwnerArgs A;
    uint32_t tc, gx, gx_fill;
    bool inst[2][2];              // per-peer form
    bool inst3[2][2][2];          // one-launch form: + [one segment of SGD / Adagrad / FTRL rows: the FAST instance]            // [one float per lane][whole-segment optimizer]
  };
  std::vector<OwnerChunk> apply_chunks;
  int64_t max_batch = 0;
  uint32_t cap = 0;             // id slots per (peer, table)
  // device memory
  int64_t* ids_send[2] = {nullptr, nullptr};   // per slot [world][ids_block]
  int64_t* ids_recv[2] = {nullptr, nullptr};
  uint32_t* slot_off[2] = {nullptr, nullptr};  // per slot [T][max_batch]
  // row buffers, each [world][rows_block]: owner side own_rows (looked-up rows out) and own_grads
  // (gradient sums in), sender side snd_rows (rows back) and snd_grads (gradient sums out).  RCCL /
  // group / identity: own_grads == own_rows, snd_grads == snd_rows (the two directions of a block
  // never overlap in time).  Peer-store transport: snd_rows and own_grads are what PEERS write, so
  // they live in this rank's window; own_rows and snd_grads stay private.
  float* own_rows = nullptr;
  float* own_grads = nullptr;
  float* snd_rows = nullptr;
  float* snd_grads = nullptr;
  // ---- owner side, between a batch's lookup and its update (ShardX, mhte_shard_kernels.h)
  OwnRec* orec = nullptr;       // [world][ids_block]
  uint32_t* oslot = nullptr;    // [world][ids_block] (world > 1)
  uint32_t* xs = nullptr;       // cross-peer scratch, [T][xcap + 1] slots (world > 1)
  uint32_t xcap = 0, xstride = 0;
  bool x_dirty = false;         // ids are registered in xs that no update has consumed
  int own_slot = -1;            // the id slot the last owner lookup served
  std::vector<uint64_t> own_epoch;   // Table::mut_epoch when that lookup ran
  bool legacy_owner = false;    // MHTE_SHARD_PER_PEER=1 (A/B), or a table has an occurrence filter: the
                                // peers' blocks are applied by one launch pair each, in rank order
  // ---- the displacement pass of an owner update rides in the NEXT owner lookup's launch (its first
  // workgroups; the lookups of a table gate on it — shard_lookup_kernel): a launch of its own cost ~6 us
  // of a ~60 us step to find, five steps in six, an empty list.  Until that lookup is enqueued the pass
  // is owed: the tables carry a hook (Table::ext_flush) that makes any other user of them run it first.
  bool fold_slow = true;        // MHTE_SHARD_FOLD_SLOW=0: always a launch of its own (A/B)
  // (claimed with an atomic exchange: the hook can fire on any thread that touches one of the tables — two
  // threads saving two tables between a backward and the next forward must not both run the pass)
  std::atomic<bool> slow_pending{false};
  int slow_slot = -1;
  hipEvent_t slow_ev = nullptr;      // recorded behind the owner update that owes the pass: a flush on another
  hipStream_t slow_stream = nullptr; // stream than the training stream is ordered behind that update
  bool slow_ev_valid = false;
  uint32_t launches = 0;        // kernel launches + exchanges enqueued by the last forward + backward
  // ---- sizing the owner's launches by what the peers actually send.  A (peer, table) block can hold the
  // whole batch; a Zipf batch fills a fifth of it, N ranks a fifth of an N-th.  Workgroups sized for the
  // capacity find no work: each still takes a slot, reads its block's count and leaves — three more
  // dispatch rounds behind the useful ones (measured by itself at one table of 65 536 ids: no change, the
  // launch's time is its dependency chain; kept for models of many tables, where the empty workgroups
  // are T times as many).  The counts
  // are on the device only, so the received blocks' headers are copied to pinned host memory behind
  // every owner lookup — nobody waits for the copy; once it has landed it sizes the LATER steps'
  // launches (grid-stride loops inside: any size is correct, a stale one merely slower).
  int64_t* h_rcnt = nullptr;    // pinned [world][hdr_words]
  hipEvent_t ev_rcnt = nullptr;
  bool rcnt_pending = false;
  std::vector<uint32_t> est_n;  // per table: ids expected in the fullest peer block (0: not known yet)
  uint32_t launches_fwd = 0;
  uint32_t* h_flags = nullptr;  // pinned, device-visible
  uint32_t* d_flags = nullptr;
  bool alias = false;           // world == 1 without a communicator
  ncclComm_t comm = nullptr;
  // MHTE_SHARD_EXACT=1: the row / gradient exchanges move only the occupied part of every (peer,
  // table) segment.  The counts are the id blocks' headers, copied to pinned host memory right
  // after the id exchange — a step ahead of their first use when the batch was prepared ahead, so
  // waiting for the copy costs nothing then.  In this mode the id blocks move exact-size as well
  // (headers first, ONE HOST WAIT per id exchange: hipEventSynchronize inside the step — the exact RCCL
  // form is host-synchronous, which also keeps its id exchange off the overlap stream).  It is the
  // DEFAULT for RCCL with whole-batch blocks (world > 1, ids_per_peer_table <= 0): fixed-size blocks that
  // can hold the whole batch would move O(world x T x batch) ids and rows per direction (26 tables x
  // 65 536 ids x 8 ranks: ~110 MB of ids each way per step).  The fixed-size form (an explicit
  // ids_per_peer_table, or MHTE_SHARD_EXACT=0) needs no host knowledge at all.  The peer-store
  // transport — the default on one node — is exact-size without any host wait.
  bool exact = false;
  // mhte_shard_step_set_exact_order: the sender's per-id gradient sums strictly in occurrence order for EVERY list
  // (lists of <= 32 are in any mode): shard_exact_sum_kernel in front of the sums' launch
  bool exact_order = false;
  int64_t* h_cnt[2] = {nullptr, nullptr};   // per slot [2 (sent | received)][world][hdr]
  hipEvent_t ev_cnt[2] = {nullptr, nullptr};
  // Round 6 — the exact form on the wire: ONE ncclSend / ncclRecv pair per peer and exchange.  The occupied
  // parts of a peer's (table) segments are packed back to back into `stage_snd` (shard_pack_kernel), the pair
  // moves the packed stream into `stage_rcv`, the receiver spreads it over its block again: pack + group +
  // unpack = 3 launches per exchange instead of world x T pairs (208 per exchange at 8 ranks x 26 tables).
  // And the id exchange's host wait has left the step: the HEADERS of batch s + 1 cross in the group of
  // batch s's gradient exchange (backward), their counts are on their way to the host from then on, and the
  // packed ids follow at the start of the next forward — by then the counts have long landed (a batch that was
  // not prepared ahead still waits, and `host_waits` counts it).
  char* stage_snd = nullptr;
  char* stage_rcv = nullptr;
  bool ids_hdr_sent[2] = {false, false};    // the slot's id HEADERS have crossed, its ids have not
  // what the last forward + backward put on the wire (mhte_shard_step_wire_stats)
  uint64_t wire_pairs = 0, wire_exchanges = 0, wire_pairs_max = 0, host_waits = 0;
  uint32_t hdr_words = 0;
  bool hdr_dirty[2] = {false, false};   // the slot's send headers hold counts
  bool disp[2] = {false, false};        // the slot's batch has been dispatched (ids exchanged)
  bool ahead = false;                   // slot cur ^ 1 holds the batch the last forward was given as next
  // ---- peer-store transport (mhte_shard_step_create_ipc; kernels: shard_push / shard_wait)
  bool ipc = false;
  bool ipc_connected = false;
  bool win_fine = false;                // the window is fine-grained device memory
  char* win = nullptr;                  // my window
  size_t win_bytes = 0;
  size_t win_off_ids[2] = {0, 0}, win_off_rows = 0, win_off_grads = 0;
  char* peer_win[kMaxShards] = {};      // every rank's window as mapped in this process
  uint32_t sig_pending[2] = {0, 0};     // channels pushed since the last sync launch (bit per channel),
                                        // per stream: [0] the caller's, [1] the step's own (overlap)
  // ---- overlap with the dense model (MHTE_SHARD_OVERLAP=1 / mhte_shard_step_set_overlap): what the
  // next batch needs and the tables do not — its run dedup, the numbering + owner packing of its
  // distinct ids, the id exchange — runs on a stream of the step's own, beside whatever the caller
  // enqueues between forward and backward (layout -> MLP -> layout gradient); the backward launch
  // then carries the gradient sums only.  The reference pipelines the same stages with its
  // prefetch queues (NT/distributed_ps_sync.py:199-203, 270-275).
  // ---- the fp16 gradient wire (MHTE_SHARD_GRAD_FP16=1 / mhte_shard_step_set_grad_bits(16)): the
  // gradient exchange moves half the bytes; a NUMERICS change the reference offers as an option
  // (NT/distributed_ps_sync.py:47,334-337), every rank must choose the same
  int grad_bits = 32;
  unsigned short* snd_grads16 = nullptr;   // sender: fp16 of snd_grads, [world][rows_block]
  unsigned short* own_grads16 = nullptr;   // owner: what arrives (peer stores: the window's gradient region)
  float* own_grads32 = nullptr;            // peer stores only: the fp32 the owner's update reads
  int overlap = 0;
  hipStream_t aux = nullptr;
  hipEvent_t ev_in = nullptr, ev_aux = nullptr;
  bool aux_pending = false;             // work on `aux` the caller's stream has not waited for yet
  bool ids_exchanged[2] = {false, false};   // the slot's id blocks have been exchanged
  // ---- direct peer stores (the default of the peer-store transport with the fp32 wire and no overlap
  // stream; MHTE_SHARD_DIRECT=0: push / sync launches as in round 4): the owner lookup writes its rows,
  // the sender's sums and numbering their gradients and ids, straight into the peers' windows, and the
  // whole protocol waits in ONE one-wavefront launch per step phase (shard_sync2_kernel)
  bool direct = false;
  unsigned long long* d_peer_win = nullptr;   // [world] window addresses, on the device
  uint32_t xa[kIpcChannels] = {};       // per channel: the exchange whose arrival was published AND awaited
  uint32_t xc[kIpcChannels] = {};       // ... whose credit (my buffer is free for it) was published and awaited
  int hdr_slot = -1;                    // id slot whose send headers go out with the next arrival
  uint32_t seq_sent[kIpcChannels] = {};               // exchanges pushed per channel
  uint32_t seq_waited[kIpcChannels][kMaxShards] = {}; // ... and waited for, per peer
  uint64_t timeout_ticks = 0;

  ~ShardStep() {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    try {
      flush_slow(nullptr);
    } catch (...) {
    }
    (void)hipDeviceSynchronize();
    if (slow_ev) (void)hipEventDestroy(slow_ev);
    if (comm) (void)Rccl::get().CommDestroy(comm);
    if (aux) (void)hipStreamDestroy(aux);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_aux) (void)hipEventDestroy(ev_aux);
    for (int p = 0; p < world && p < kMaxShards; ++p)
      if (peer_win[p] && p != rank) (void)hipIpcCloseMemHandle(peer_win[p]);
    for (int s = 0; s < 2; ++s) {
      if (ids_send[s]) (void)hipFree(ids_send[s]);
      if (ids_recv[s] && !alias && !ipc) (void)hipFree(ids_recv[s]);
      if (slot_off[s]) (void)hipFree(slot_off[s]);
    }
    if (d_tab) (void)hipFree(d_tab);
    if (d_peer_win) (void)hipFree(d_peer_win);
    if (h_rcnt) (void)hipHostFree(h_rcnt);
    if (ev_rcnt) (void)hipEventDestroy(ev_rcnt);
    if (orec) (void)hipFree(orec);
    if (oslot) (void)hipFree(oslot);
    if (xs) (void)hipFree(xs);
    if (own_rows) (void)hipFree(own_rows);
    if (snd_rows && !alias && !ipc) (void)hipFree(snd_rows);
    if (ipc && snd_grads) (void)hipFree(snd_grads);
    if (snd_grads16) (void)hipFree(snd_grads16);
    if (own_grads16 && !ipc) (void)hipFree(own_grads16);
    if (own_grads32) (void)hipFree(own_grads32);
    if (win) (void)hipFree(win);
    if (h_flags) (void)hipHostFree(h_flags);
    for (int s = 0; s < 2; ++s) {
      if (h_cnt[s]) (void)hipHostFree(h_cnt[s]);
      if (ev_cnt[s]) (void)hipEventDestroy(ev_cnt[s]);
    }
    if (stage_snd) (void)hipFree(stage_snd);
    if (stage_rcv) (void)hipFree(stage_rcv);
  }

  void init(mhte_multi_table* m, int64_t mb, int rank_, int world_, int64_t ids_per_peer_table,
            const void* unique_id, bool ipc_ = false) {
    mt = m;
    ipc = ipc_;
    device = m->device;
    rank = rank_;
    world = world_;
    max_batch = mb;
    T = uint32_t(m->tables.size());
    if (world < 1 || world > kMaxShards || rank < 0 || rank >= world)
      throw Error(MHTE_INVALID_ARGUMENT, "shard step: rank / world out of range (world <= " +
                                             std::to_string(kMaxShards) + ")");
    // (any number of tables: the launches take kMaxStepTables of them each — table_chunks)
    if (T < 1 || uint64_t(T) * uint64_t(world) > 65535ull)
      throw Error(MHTE_INVALID_ARGUMENT, "shard step: tables x world must be 1..65535");
    if (!seg_shapes_ok(m))
      throw Error(MHTE_INVALID_ARGUMENT, "shard step: every table needs rows of whole float4s up to "
                                         "256 floats or of any layout up to 64");
    // (the sender side — dedup, numbering, scatter, gradient sums — does not look at the optimizer: a
    // GroupAdaGrad table is fine here, its owner applies it with the whole-segment instance)
    ms.init(m, mb, /*sender_roles_only=*/true);
    // default: a (peer, table) block can hold the whole batch, so no step can overflow one (the
    // reference's all-to-all is variable-sized and never drops an id).  A smaller capacity is the
    // caller's explicit choice (fixed-size RCCL blocks that cross the links whole).
    int64_t c = ids_per_peer_table > 0 ? ids_per_peer_table : mb;
    c = std::min<int64_t>(c, mb);
    cap = uint32_t((c + 3) & ~int64_t(3));
    geo.world = uint32_t(world);
    geo.T = T;
    const uint32_t hdr = (T + 7u) & ~7u;
    hdr_words = hdr;
    tab.resize(T);
    uint64_t idw = hdr, rw = 0;
    for (uint32_t t = 0; t < T; ++t) {
      tab[t].cap = cap;
      tab[t].dim = m->tables[t]->dim;
      tab[t].id_off = uint32_t(idw);
      tab[t].row_off = uint32_t(rw);
      idw += cap;
      rw += uint64_t(cap) * tab[t].dim;
    }
    if (rw * uint64_t(world) >= 0xffffffffull || idw * uint64_t(world) >= 0xffffffffull)
      throw Error(MHTE_INVALID_ARGUMENT, "shard step: peer blocks exceed 2^32 floats; lower "
                                         "ids_per_peer_table");
    geo.ids_block = uint32_t((idw + 1) & ~uint64_t(1));
    geo.rows_block = uint32_t(rw);
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_tab), sizeof(ShardTab) * T));
    HIP_OK(hipMemcpy(d_tab, tab.data(), sizeof(ShardTab) * T, hipMemcpyHostToDevice));
    alias = world == 1 && unique_id == nullptr && !ipc;
    const size_t ib = size_t(geo.ids_block) * world * sizeof(int64_t);
    const size_t rb = size_t(geo.rows_block) * world * sizeof(float);
    if (ipc) alloc_window(ib, rb);
    for (int s = 0; s < 2; ++s) {
      HIP_OK(hipMalloc(&ids_send[s], ib));
      HIP_OK(hipMemset(ids_send[s], 0, ib));
      if (alias) {
        ids_recv[s] = ids_send[s];
      } else if (ipc) {
        ids_recv[s] = reinterpret_cast<int64_t*>(win + win_off_ids[s]);
      } else {
        HIP_OK(hipMalloc(&ids_recv[s], ib));
        HIP_OK(hipMemset(ids_recv[s], 0, ib));
      }
      HIP_OK(hipMalloc(&slot_off[s], size_t(T) * size_t(mb) * sizeof(uint32_t)));
    }
    // what the owner keeps per received id between lookup and update; world > 1: the cross-peer scratch
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&orec), size_t(geo.ids_block) * world * sizeof(OwnRec)));
    HIP_OK(hipMemset(orec, 0xff, size_t(geo.ids_block) * world * sizeof(OwnRec)));
    own_epoch.assign(T, 0);
    est_n.assign(T, 0);
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&h_rcnt), size_t(world) * hdr * sizeof(int64_t), hipHostMallocDefault));
    memset(h_rcnt, 0, size_t(world) * hdr * sizeof(int64_t));
    HIP_OK(hipEventCreateWithFlags(&ev_rcnt, hipEventDisableTiming));
    if (world > 1) {
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&oslot), size_t(geo.ids_block) * world * sizeof(uint32_t)));
      HIP_OK(hipMemset(oslot, 0, size_t(geo.ids_block) * world * sizeof(uint32_t)));
      xcap = 1u << std::max<uint32_t>(6, ceil_log2(uint64_t(2) * uint64_t(cap) * uint64_t(world)));
      xstride = (kXKeyWords + uint32_t(world) + 3u) & ~3u;
      const size_t words = size_t(T) * (size_t(xcap) + 1) * xstride;
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&xs), words * sizeof(uint32_t)));
      clear_x(nullptr);
    }
    if (const char* e = getenv("MHTE_SHARD_PER_PEER")) legacy_owner = atoi(e) != 0;
    if (const char* e = getenv("MHTE_SHARD_FOLD_SLOW")) fold_slow = atoi(e) != 0;
    if (const char* e = getenv("MHTE_SHARD_FUSE_SCATTER")) fuse_scatter = atoi(e) != 0;
    HIP_OK(hipMalloc(&own_rows, rb + 64));
    if (ipc) {
      snd_rows = reinterpret_cast<float*>(win + win_off_rows);
      own_grads = reinterpret_cast<float*>(win + win_off_grads);
      HIP_OK(hipMalloc(&snd_grads, rb + 64));
    } else {
      if (alias) snd_rows = own_rows;
      else HIP_OK(hipMalloc(&snd_rows, rb + 64));
      own_grads = own_rows;
      snd_grads = snd_rows;
    }
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&h_flags), 64, hipHostMallocMapped));
    memset(h_flags, 0, 64);
    HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_flags), h_flags, 0));
    // RCCL with whole-batch blocks: exact-size exchanges unless the caller insists (a fixed-size
    // exchange of blocks that can hold the whole batch would move world x the batch per direction)
    exact = unique_id != nullptr && world > 1 && ids_per_peer_table <= 0;
    if (const char* e = getenv("MHTE_SHARD_EXACT")) exact = atoi(e) != 0;
    exact = exact && !alias && !ipc;
    if (exact)
      for (int s = 0; s < 2; ++s) {
        HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&h_cnt[s]), size_t(2) * world * hdr * sizeof(int64_t),
                             hipHostMallocDefault));
        memset(h_cnt[s], 0, size_t(2) * world * hdr * sizeof(int64_t));
        HIP_OK(hipEventCreateWithFlags(&ev_cnt[s], hipEventDisableTiming));
      }
    if (exact) {
      const size_t sb = std::max(ib, rb) + 64;
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&stage_snd), sb));
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&stage_rcv), sb));
    }
    if (const char* e = getenv("MHTE_SHARD_OVERLAP")) set_overlap(atoi(e));
    // (a job-wide setting: the world-1 identity step has no wire to narrow and ignores it)
    if (const char* e = getenv("MHTE_SHARD_GRAD_FP16"))
      if (!alias) set_grad_bits(atoi(e) != 0 ? 16 : 32);
    if (unique_id) {
      Rccl& R = Rccl::get();
      ncclUniqueId id;
      memcpy(&id, unique_id, sizeof(id));
      R.ok(R.CommInitRank(&comm, world, id, rank), "CommInitRank");
    }
    HIP_OK(hipDeviceSynchronize());
  }

  bool local_group_member() const { return world > 1 && comm == nullptr && !ipc; }

  // (the wire format and the pipeline mode travel in the window handle: every rank is checked against
  // every other at connect — so they cannot change behind a connected window of a world > 1)
  void refuse_after_connect(bool changes, const char* what) const {
    if (changes && ipc && ipc_connected && (world > 1 || direct))
      throw Error(MHTE_FAILED_PRECONDITION,
                  std::string("shard step: ") + what + " must be chosen before the window handles are exchanged "
                  "(ShardedMultiStep(grad_fp16= / overlap=), MHTE_SHARD_GRAD_FP16 / MHTE_SHARD_OVERLAP, or the "
                  "setter between mhte_shard_step_create_ipc and mhte_shard_step_ipc_handle): the ranks agree on "
                  "it at connect");
  }
  void set_grad_bits(int bits) {
    if (bits != 16 && bits != 32) throw Error(MHTE_INVALID_ARGUMENT, "shard step: gradient wire is 32 or 16 bits");
    refuse_after_connect(bits != grad_bits, "the gradient wire (fp16 / fp32)");
    if (bits == 16 && alias)
      throw Error(MHTE_INVALID_ARGUMENT, "shard step: the identity exchange (world 1) has no wire to narrow");
    if (bits == 16 && !snd_grads16) {
      const size_t hb = size_t(geo.rows_block) * world * 2 + 64;
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&snd_grads16), hb));
      if (ipc) {
        own_grads16 = reinterpret_cast<unsigned short*>(win + win_off_grads);
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&own_grads32), size_t(geo.rows_block) * world * 4 + 64));
      } else {
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&own_grads16), hb));
      }
    }
    grad_bits = bits;
  }
  // the fp32 buffer the owner's update reads
  float* apply_grads() const { return (grad_bits == 16 && ipc) ? own_grads32 : own_grads; }
  template <bool NARROW>
  void cvt(const int64_t* counts, const void* src, void* dst, int peer_lo, int peer_n, hipStream_t st) {
    ShardCvtArgs A{};
    A.counts = counts;
    A.src = src;
    A.dst = dst;
    A.geo = geo;
    A.peer_lo = uint32_t(peer_lo);
    A.peer_n = uint32_t(peer_n);
    A.tab = d_tab;
    uint32_t gx = 1;
    for (uint32_t t = 0; t < T; ++t) gx = std::max(gx, (cap * tab[t].dim / 4u + 1023u) / 1024u);
    gx = std::min<uint32_t>(gx, std::max<uint32_t>(4, uint32_t(ms.num_cus) * 4 / (uint32_t(peer_n) * T)));
    shard_cvt_kernel<NARROW><<<dim3(gx, uint32_t(peer_n) * T), 256, 0, st>>>(A);
    HIP_OK(hipGetLastError());
  }

  void set_overlap(int mode) {
    refuse_after_connect((mode != 0 ? 1 : 0) != overlap, "the overlap mode");
    overlap = mode != 0 ? 1 : 0;
    if (overlap && !aux) {
      // (lowest priority: the dense model's GEMMs on the caller's stream go first where both want a CU)
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, least) != hipSuccess) {
        (void)hipGetLastError();
        HIP_OK(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
      }
      HIP_OK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&ev_aux, hipEventDisableTiming));
    }
  }
  // the caller's stream waits for what was enqueued on the step's own
  void join_aux(hipStream_t st) {
    if (!aux_pending) return;
    HIP_OK(hipStreamWaitEvent(st, ev_aux, 0));
    aux_pending = false;
  }
  // the next batch on the step's own stream: dedup, numbering + packing, id exchange where the
  // transport allows it there (peer stores; RCCL keeps one stream per communicator)
  void prepare_next_on_aux(const int64_t* ids, const int64_t* split, int slot, hipStream_t st) {
    HIP_OK(hipEventRecord(ev_in, st));       // the ids are ready, the slot's last users are done
    HIP_OK(hipStreamWaitEvent(aux, ev_in, 0));
    dedup(ids, split, slot, aux);
    build_and_sum(slot, -1, nullptr, aux);
    ids_exchanged[slot] = false;
    ids_hdr_sent[slot] = false;
    if (alias) {
      ids_exchanged[slot] = true;
    } else if (ipc) {
      exchange_ipc(kXIds, slot, aux);
      flush_signals(aux);
      ids_exchanged[slot] = true;
    }
    HIP_OK(hipEventRecord(ev_aux, aux));
    aux_pending = true;
  }

  // ---- peer-store transport: the window -------------------------------------------------------------
  void alloc_window(size_t ib, size_t rb) {
    auto up = [](size_t x) { return (x + 4095) & ~size_t(4095); };
    size_t off = up(kIpcFlagBytes);
    for (int s = 0; s < 2; ++s) {
      win_off_ids[s] = off;
      off += up(ib);
    }
    win_off_rows = off;
    off += up(rb + 64);
    win_off_grads = off;
    off += up(rb + 64);
    win_bytes = off;
    // fine-grained device memory: written by other agents / processes while this one reads it
    // (MHTE_SHARD_WINDOW=coarse: plain hipMalloc, for A/B runs)
    const char* e = getenv("MHTE_SHARD_WINDOW");
    const bool want_fine = !(e && std::string(e) == "coarse");
    void* w = nullptr;
    if (want_fine && hipExtMallocWithFlags(&w, win_bytes, hipDeviceMallocFinegrained) == hipSuccess) {
      win_fine = true;
    } else {
      (void)hipGetLastError();
      HIP_OK(hipMalloc(&w, win_bytes));
    }
    win = static_cast<char*>(w);
    HIP_OK(hipMemset(win, 0, win_bytes));
    double ms = 30000.0;
    if (const char* t = getenv("MHTE_SHARD_TIMEOUT_MS")) ms = std::max(1.0, atof(t));
    timeout_ticks = uint64_t(ms * 1e5);   // wall_clock64: 100 MHz
    peer_win[rank] = win;
  }

  struct IpcBlob {               // what mhte_shard_step_ipc_handle hands the launcher (128 bytes)
    hipIpcMemHandle_t h;         // 64
    uint64_t win_bytes;
    uint32_t magic, rank, world, T, cap, ids_block, rows_block;
    int32_t pid;
    uint32_t grad_bits, overlap;  // the wire format and the pipeline mode: every rank the same
    uint32_t direct;              // ... and the form of the exchanges (direct stores / push launches)
    char pad[128 - 64 - 8 - 7 * 4 - 4 - 8 - 4];
  };
  static_assert(sizeof(IpcBlob) == 128, "ipc handle blob");
  static constexpr uint32_t kIpcMagic = 0x6d687431u;

  void ipc_handle(void* out128) {
    if (!ipc) throw Error(MHTE_FAILED_PRECONDITION, "shard step: not created with the peer-store transport");
    IpcBlob b{};
    HIP_OK(hipIpcGetMemHandle(&b.h, win));
    b.win_bytes = win_bytes;
    b.magic = kIpcMagic;
    b.rank = uint32_t(rank);
    b.world = uint32_t(world);
    b.T = T;
    b.cap = cap;
    b.ids_block = geo.ids_block;
    b.rows_block = geo.rows_block;
    b.pid = int32_t(getpid());
    b.grad_bits = uint32_t(grad_bits);   // (set_grad_bits / set_overlap come BEFORE the handle is taken)
    b.overlap = uint32_t(overlap);
    b.direct = want_direct() ? 1u : 0u;
    memcpy(out128, &b, sizeof(b));
  }

  // handles: world x 128 bytes, rank-major (every rank's mhte_shard_step_ipc_handle output)
  void ipc_connect(const void* handles) {
    if (!ipc) throw Error(MHTE_FAILED_PRECONDITION, "shard step: not created with the peer-store transport");
    if (ipc_connected) throw Error(MHTE_FAILED_PRECONDITION, "shard step: already connected");
    const char* hb = static_cast<const char*>(handles);
    for (int p = 0; p < world; ++p) {
      IpcBlob b;
      memcpy(&b, hb + size_t(p) * sizeof(IpcBlob), sizeof(b));
      if (b.magic != kIpcMagic || b.rank != uint32_t(p) || b.world != uint32_t(world))
        throw Error(MHTE_INVALID_ARGUMENT, "shard step connect: handle " + std::to_string(p) +
                                               " is not rank " + std::to_string(p) + " of this world");
      if (b.T != T || b.cap != cap || b.ids_block != geo.ids_block || b.rows_block != geo.rows_block ||
          b.win_bytes != win_bytes)
        throw Error(MHTE_INVALID_ARGUMENT, "shard step connect: rank " + std::to_string(p) +
                                               " was created with other tables or capacities");
      // (fp16 against fp32 gradient blocks would corrupt silently; overlap on one side only ends in a
      // peer timeout)
      if (b.grad_bits != uint32_t(grad_bits) || b.overlap != uint32_t(overlap))
        throw Error(MHTE_INVALID_ARGUMENT,
                    "shard step connect: rank " + std::to_string(p) + " runs gradient wire " +
                        std::to_string(b.grad_bits) + " bits / overlap " + std::to_string(b.overlap) +
                        ", this rank " + std::to_string(grad_bits) + " / " + std::to_string(overlap) +
                        " (mhte_shard_step_set_grad_bits / _set_overlap on every rank, before the handles are taken)");
      if (b.direct != (want_direct() ? 1u : 0u))
        throw Error(MHTE_INVALID_ARGUMENT, "shard step connect: rank " + std::to_string(p) + " runs the " +
                                               (b.direct ? "direct-store" : "push-launch") + " form of the exchanges, this "
                                               "rank the other (MHTE_SHARD_DIRECT must agree)");
      if (p == rank) continue;
      void* m = nullptr;
      hipError_t e = hipIpcOpenMemHandle(&m, b.h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        throw Error(MHTE_UNAVAILABLE, std::string("shard step connect: hipIpcOpenMemHandle of rank ") +
                                          std::to_string(p) + "'s window: " + hipGetErrorString(e));
      }
      peer_win[p] = static_cast<char*>(m);
    }
    ipc_connected = true;
    direct = want_direct();
    if (direct) {
      std::vector<unsigned long long> w(size_t(world), 0ull);
      for (int p = 0; p < world; ++p) w[size_t(p)] = reinterpret_cast<unsigned long long>(peer_win[p]);
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_peer_win), sizeof(unsigned long long) * size_t(world)));
      HIP_OK(hipMemcpy(d_peer_win, w.data(), sizeof(unsigned long long) * size_t(world), hipMemcpyHostToDevice));
    }
  }
  bool want_direct() const {
    static const bool off = getenv("MHTE_SHARD_DIRECT") && atoi(getenv("MHTE_SHARD_DIRECT")) == 0;
    return ipc && !off && grad_bits == 32 && overlap == 0;
  }

  static uint32_t chan_of(int kind, int slot) {
    return kind == kXIds ? uint32_t(slot ? kChIds1 : kChIds0) : kind == kXRows ? uint32_t(kChRows) : uint32_t(kChGrads);
  }

  void push(uint32_t chan, const void* src, const int64_t* counts, size_t dst_off, bool ids, hipStream_t st,
            bool half = false) {
    if (!ipc_connected) throw Error(MHTE_FAILED_PRECONDITION, "shard step: mhte_shard_step_ipc_connect first");
    ShardPushArgs A{};
    for (int p = 0; p < world; ++p) A.win[p] = peer_win[p];
    A.src = static_cast<const char*>(src);
    A.counts = counts;
    A.dst_off = dst_off;
    A.flags = d_flags;
    A.timeout_ticks = timeout_ticks;
    A.geo = geo;
    A.rank = uint32_t(rank);
    A.chan = chan;
    A.seq = ++seq_sent[chan];
    A.ids = ids ? 1u : 0u;
    A.half = half ? 1u : 0u;
    A.tab = d_tab;
    // enough workgroups per peer to keep a link (or the local HBM) busy, few enough that a waiting
    // launch never fills the chip (another process may share the device)
    const size_t blk = ids ? size_t(geo.ids_block) * 8 : size_t(geo.rows_block) * 4;
    uint32_t gx = src ? uint32_t(std::min<size_t>(std::max<size_t>(1, blk / (256 * 16 * 4)),
                                                  std::max<size_t>(1, size_t(ms.num_cus) * 2 / size_t(world))))
                      : 1u;
    LAUNCH_HOT(kTagShardPush, shard_push_kernel, dim3(gx, uint32_t(world)), 256, st, A);
    ++launches;
    HIP_OK(hipGetLastError());
    sig_pending[st == aux && aux ? 1 : 0] |= 1u << chan;
  }

  // one launch: publish the arrival of every push since the last sync, then (wait_chan <
  // kIpcChannels) hold the stream until peers [lo, hi) have published `wait_chan`'s latest exchange
  void sync(uint32_t wait_chan, int lo, int hi, hipStream_t st) {
    uint32_t& pend = sig_pending[st == aux && aux ? 1 : 0];   // (a push is published on ITS stream)
    if (!pend && wait_chan >= uint32_t(kIpcChannels)) return;
    ShardSyncArgs A{};
    for (int p = 0; p < world; ++p) A.win[p] = peer_win[p];
    A.flags = d_flags;
    A.timeout_ticks = timeout_ticks;
    A.rank = uint32_t(rank);
    A.world = uint32_t(world);
    for (uint32_t c = 0; c < uint32_t(kIpcChannels); ++c)
      if (pend & (1u << c)) {
        A.sig_chan[A.n_sig] = c;
        A.sig_seq[A.n_sig++] = seq_sent[c];
      }
    pend = 0;
    A.wait_chan = wait_chan;
    if (wait_chan < uint32_t(kIpcChannels)) {
      A.wait_seq = seq_sent[wait_chan];
      A.lo = uint32_t(lo);
      A.hi = uint32_t(hi);
    }
    LAUNCH_HOT(kTagShardWait, shard_sync_kernel, 1, 64, st, A);
    ++launches;
    HIP_OK(hipGetLastError());
  }
  // at the end of an API call: nothing this rank owes its peers stays unpublished
  void flush_signals(hipStream_t st) {
    if (direct) {
      sync_point(0u, st);
      return;
    }
    if (ipc && sig_pending[st == aux && aux ? 1 : 0]) sync(uint32_t(kIpcChannels), 0, 0, st);
  }

  // ---- direct mode: the one-wavefront sync point (shard_sync2_kernel).  Publishes the arrival of every
  // exchange written since the last one (+ the headers of an id exchange) and the credits in `credit_mask`
  // that are not out yet; waits for the same from every peer.
  void sync_point(uint32_t credit_mask, hipStream_t st) {
    ShardSync2Args A{};
    for (int p = 0; p < world; ++p) A.win[p] = peer_win[p];
    A.flags = d_flags;
    A.timeout_ticks = timeout_ticks;
    A.rank = uint32_t(rank);
    A.world = uint32_t(world);
    A.hdr_words = hdr_words;
    A.ids_block = geo.ids_block;
    for (uint32_t ch = 0; ch < uint32_t(kChTest); ++ch) {
      if (xa[ch] != seq_sent[ch]) {
        A.arr_chan[A.n_arr] = ch;
        A.arr_seq[A.n_arr++] = seq_sent[ch];
        xa[ch] = seq_sent[ch];
        if (ch <= uint32_t(kChIds1) && hdr_slot == int(ch)) {   // (kChIds0 / kChIds1 = id slot 0 / 1)
          A.hdr_src = ids_send[hdr_slot];
          A.hdr_dst_off = win_off_ids[hdr_slot] + size_t(rank) * size_t(geo.ids_block) * 8;
          hdr_slot = -1;
        }
      }
      if (((credit_mask >> ch) & 1u) && xc[ch] != seq_sent[ch] + 1u) {
        A.cred_chan[A.n_cred] = ch;
        A.cred_seq[A.n_cred++] = seq_sent[ch] + 1u;
        xc[ch] = seq_sent[ch] + 1u;
      }
    }
    if (!A.n_arr && !A.n_cred) return;
    LAUNCH_HOT(kTagShardWait, shard_sync2_kernel, 1, 64, st, A);
    HIP_OK(hipGetLastError());
    ++launches;
  }
  // before a launch that stores exchange seq_sent[ch] + 1 of the channels in `mask` into the peers' windows
  void require_credits(uint32_t mask, hipStream_t st) {
    uint32_t missing = 0;
    for (uint32_t ch = 0; ch < uint32_t(kChTest); ++ch)
      if (((mask >> ch) & 1u) && xc[ch] != seq_sent[ch] + 1u) missing |= 1u << ch;
    if (missing) sync_point(missing, st);
  }
  // before a consumer of the channels' latest exchanges (more_credits: what is free to hand out at this point)
  void require_arrived(uint32_t mask, uint32_t more_credits, hipStream_t st) {
    for (uint32_t ch = 0; ch < uint32_t(kChTest); ++ch)
      if (((mask >> ch) & 1u) && xa[ch] != seq_sent[ch]) {
        sync_point(more_credits, st);
        return;
      }
  }
  static uint32_t chbit(int kind, int slot) { return 1u << chan_of(kind, slot); }

  void exchange_ipc(int kind, int slot, hipStream_t st) {
    if (direct) return;   // (the producer launch stored into the peers' windows itself)
    const uint32_t ch = chan_of(kind, slot);
    if (kind == kXIds)
      push(ch, ids_send[slot], ids_send[slot], win_off_ids[slot], true, st);
    else if (kind == kXRows)   // rows of the ids every peer sent me: sized by the received headers
      push(ch, own_rows, ids_recv[slot], win_off_rows, false, st);
    else if (grad_bits == 16)  // gradient sums of the ids I sent every peer, as fp16
      push(ch, snd_grads16, ids_send[slot], win_off_grads, false, st, true);
    else
      push(ch, snd_grads, ids_send[slot], win_off_grads, false, st);
  }

  // before a consumer of what peers [lo, hi) sent on (kind, slot): hold the stream until it landed
  void wait_arrived(int kind, int slot, int lo, int hi, hipStream_t st, uint32_t more_credits = 0u) {
    if (!ipc) return;
    if (direct) {
      require_arrived(chbit(kind, slot), more_credits, st);
      return;
    }
    const uint32_t ch = chan_of(kind, slot);
    int a = hi, b = lo;
    for (int p = lo; p < hi; ++p)
      if (seq_waited[ch][p] != seq_sent[ch]) {
        a = std::min(a, p);
        b = std::max(b, p + 1);
      }
    if (a >= b) return;
    sync(ch, a, b, st);
    for (int p = a; p < b; ++p) seq_waited[ch][p] = seq_sent[ch];
  }

  // a data-less round trip with every peer: proves the windows are mapped and the flags travel
  // Collective.  Three rounds of: credits (a data-less push: every peer's test region is free) ->
  // pattern into every peer's window -> arrival published and awaited -> check of what the peers
  // wrote here (shard_selftest_kernel).  A transport that loses, delays or caches the peers' stores
  // fails HERE — and the caller falls back to RCCL — instead of training on stale rows.
  void ipc_selftest(hipStream_t st) {
    ShardSelftestArgs A{};
    for (int p = 0; p < world; ++p) A.win[p] = peer_win[p];
    A.off = win_off_grads;
    A.blk = size_t(geo.rows_block) * 4;
    A.flags = d_flags;
    A.n16 = uint32_t(std::min<size_t>(4096, A.blk) / 16);
    A.rank = uint32_t(rank);
    if (const char* e = getenv("MHTE_SHARD_SELFTEST_CORRUPT")) A.corrupt = uint32_t(atoi(e));
    for (uint32_t round = 0; round < 3; ++round) {
      push(kChTest, nullptr, ids_send[0], 0, false, st);
      A.round = round;
      if (A.n16) {
        A.check = 0;
        shard_selftest_kernel<<<uint32_t(world), 256, 0, st>>>(A);
        HIP_OK(hipGetLastError());
      }
      sync(uint32_t(kChTest), 0, world, st);
      if (A.n16) {
        A.check = 1;
        shard_selftest_kernel<<<uint32_t(world), 256, 0, st>>>(A);
        HIP_OK(hipGetLastError());
      }
    }
    HIP_OK(hipStreamSynchronize(st));
    check_flags();
  }

  void check_flags() {
    const uint32_t f = *reinterpret_cast<volatile uint32_t*>(h_flags);
    if (f & kShardPeerTimeout) {
      *reinterpret_cast<volatile uint32_t*>(h_flags) = 0;
      throw Error(MHTE_UNAVAILABLE, "shard step: a peer did not take part in an exchange within "
                                    "MHTE_SHARD_TIMEOUT_MS (its process is gone, or the ranks' calls "
                                    "are out of step); the step's results are not valid");
    }
    if (f & kShardSelftestBad) {
      *reinterpret_cast<volatile uint32_t*>(h_flags) = 0;
      throw Error(MHTE_UNAVAILABLE, "shard step: the peer-store self test read other data than a peer "
                                    "wrote (stores into a mapped window are not arriving intact on this "
                                    "system); use the RCCL transport");
    }
    if (f) {
      *reinterpret_cast<volatile uint32_t*>(h_flags) = 0;
      throw Error(MHTE_RESOURCE_EXHAUSTED,
                  "shard step: a table sent more than ids_per_peer_table = " + std::to_string(cap) +
                      " distinct ids to one peer in a step; their rows were zeros and their gradients "
                      "dropped.  Create the step with a larger capacity");
    }
  }

  void prepare(hipStream_t st, bool keep_owed_pass = false) {
    check_flags();
    if (aux_pending)   // (descriptors are about to be re-uploaded: nothing of ours may still read them)
      for (uint32_t t = 0; t < T; ++t)
        if (ms.st_version[t] != mt->tables[t]->view_version) {
          HIP_OK(hipStreamSynchronize(aux));
          break;
        }
    for (auto& tb : mt->tables) tb->finish_pending(st, keep_owed_pass);
    sync_views(mt, st);
    ms.sync_static(st);
  }

  // tables [t0, t0 + tc) of one launch
  uint32_t chunk_tables(uint32_t t0) const { return std::min<uint32_t>(uint32_t(kMaxStepTables), T - t0); }
  void fill_tabs(ShardTab* dst, uint32_t t0, uint32_t tc) const {
    for (uint32_t i = 0; i < tc; ++i) dst[i] = tab[t0 + i];
  }

  // ---- sender: run dedup of (ids, split) into `slot` (stage 1)
  void dedup(const int64_t* ids, const int64_t* split, int slot, hipStream_t st) {
    if (ms.stage[slot] == 1) ms.clear_slots(1u << slot, st);
    for (uint32_t t = 0; t < T; ++t) ms.n_slot[slot][t] = uint32_t(split[t + 1] - split[t]);
    ms.has_hints[slot] = false;
    ms.launch_dedup(ids, split, slot, 0, st);
    launches += (T + uint32_t(kMaxStepTables) - 1) / uint32_t(kMaxStepTables);
    ms.stage[slot] = 1;
    disp[slot] = false;
    ids_exchanged[slot] = false;
    ids_hdr_sent[slot] = false;
  }

  // -> gt_all[T]: the workgroup shares are of the whole model (every chunk's launches run together)
  void gather_tabs(int slot) const {
    ShardGatherTab* gt = (gt_all.assign(T, ShardGatherTab{}), gt_all.data());
    uint32_t active = 0;
    for (uint32_t t = 0; t < T; ++t) active += ms.n_slot[slot][t] ? 1u : 0u;
    const uint32_t budget = uint32_t(kBwdBlocksPerCu * ms.num_cus);
    const uint32_t share = std::max<uint32_t>(32, budget * (active > 1 ? ms.ovs : 1u) / std::max(1u, active));
    int64_t off = 0;
    for (uint32_t t = 0; t < T; ++t) {
      ShardGatherTab& g = gt[t];
      const uint32_t n = ms.n_slot[slot][t];
      g.n = n;
      g.io_off = uint32_t(off);
      off += int64_t(n) * tab[t].dim;
      g.nblk_items = g.nblk_ids = 0;
      // (rows in the wire blocks sit at row_off + slot * dim: on 16-byte boundaries exactly when the
      // row is whole float4s, which vec_ok says; the flat buffer's slice may still be off)
      g.gv = shape_code(*ms.mt->tables[t], uint64_t(g.io_off));
      if (!n) continue;
      const uint32_t groups_per_wg = 256u / shape_lanes(g.gv);
      g.nblk_items = std::min<uint32_t>(DedupWs::max_items(n),
                                        std::min<uint32_t>(uint32_t(ms.num_cus) * 10 / 8,
                                                           std::max<uint32_t>(8, share / 4)));
      const uint32_t need = (n + groups_per_wg - 1) / groups_per_wg;
      g.nblk_ids = std::max<uint32_t>(1, std::min(need, share));
    }
    if (uint64_t(off) > 0xffffffffull)
      throw Error(MHTE_INVALID_ARGUMENT, "shard step: flat buffer exceeds 2^32 floats");
  }

  // ---- sender: [gradient sums of the batch in sum_slot -> row slots] | [numbering + owner packing
  // of the batch deduplicated into build_slot]; either may be absent (-1)
  void build_and_sum(int build_slot, int sum_slot, const float* grads, hipStream_t st) {
    ShardBuildArgs A{};
    A.st = ConstStatics(ms.d_st);
    A.geo = geo;
    A.flags = d_flags;
    A.n_max = uint32_t(max_batch);
    gt_all.assign(T, ShardGatherTab{});
    // (an owed displacement pass reads the id and gradient blocks of its update and clears send headers: it
    // runs before this rank tells its peers that those buffers are free, and before this numbering)
    if (build_slot >= 0) flush_slow(st);
    if (direct) {   // sums and ids go straight into the owners' windows
      require_credits((sum_slot >= 0 ? chbit(kXGrads, sum_slot) : 0u) | (build_slot >= 0 ? chbit(kXIds, build_slot) : 0u), st);
      A.peer_win = d_peer_win;
      A.peer_grads_off = win_off_grads + size_t(rank) * size_t(geo.rows_block) * sizeof(float);
      if (build_slot >= 0) A.peer_ids_off = win_off_ids[build_slot] + size_t(rank) * size_t(geo.ids_block) * sizeof(int64_t);
    }
    if (sum_slot >= 0) {
      A.grads = grads;
      A.rows_out = snd_grads;
      A.slot_off = slot_off[sum_slot];
      A.slot = uint32_t(sum_slot);
      gather_tabs(sum_slot);
    }   // (no sums: shape code 0, the numbering alone runs in the float4 instance)
    if (build_slot >= 0) {
      flush_slow(st);   // (the owed pass clears send headers: it must not follow this numbering)
      if (hdr_dirty[build_slot])   // (a batch that was packed and never trained)
        HIP_OK(hipMemset2DAsync(ids_send[build_slot], size_t(geo.ids_block) * 8, 0, size_t(T) * 8,
                                size_t(world), st));
      A.send_ids = ids_send[build_slot];
      A.slot_off_build = slot_off[build_slot];
      A.build_slot = uint32_t(build_slot);
      hdr_dirty[build_slot] = true;
      ms.stage[build_slot] = 2;
      disp[build_slot] = true;
    }
    for (uint32_t t0 = 0; t0 < T; t0 += uint32_t(kMaxStepTables)) {
      const uint32_t tc = chunk_tables(t0);
      A.t0 = t0;
      fill_tabs(A.tab, t0, tc);
      uint32_t gx = 0;   // (the grid is the per-table maximum of build + sum blocks)
      bool w4 = false, w1 = false;
      for (uint32_t i = 0; i < tc; ++i) {
        const uint32_t t = t0 + i;
        A.gt[i] = gt_all[t];
        A.n_build[i] = build_slot >= 0 ? ms.n_slot[build_slot][t] : 0u;
        gx = std::max(gx, (A.n_build[i] ? ms.h_st[t].nblk_build : 0u) + A.gt[i].nblk_items + A.gt[i].nblk_ids);
        // (one instance of the kernel per lane width among the tables: mhte_mstep_kernels.h MHTE_SWITCH_G)
        ((A.gt[i].gv & 1u) ? w1 : w4) = true;
      }
      if (!gx) continue;
      A.exact = (exact_order && sum_slot >= 0) ? 1u : 0u;
      if (A.exact) {   // (one 135-KB workgroup per CU: the chip's CUs dealt over the launch's tables)
        shard_exact_sum_kernel<<<dim3(std::max<uint32_t>(4, uint32_t(ms.num_cus) / tc), tc), kExactThreads, 0, st>>>(A);
        ++launches;
      }
      if (w4) LAUNCH_HOT(kTagShardBuild, shard_build_kernel<4>, dim3(gx, tc), 256, st, A);
      if (w1) LAUNCH_HOT(kTagShardBuild, shard_build_kernel<1>, dim3(gx, tc), 256, st, A);
      launches += (w4 ? 1u : 0u) + (w1 ? 1u : 0u);
      HIP_OK(hipGetLastError());
    }
    if (direct) {   // (the exchanges these launches were: every rank counts them alike, whatever it sent)
      if (sum_slot >= 0) ++seq_sent[kChGrads];
      if (build_slot >= 0) {
        ++seq_sent[chan_of(kXIds, build_slot)];
        hdr_slot = build_slot;
      }
    }
  }

  void scatter(float* out, int slot, hipStream_t st) {
    // (direct mode: at this point the gradient buffer and the other slot's id buffer are free for the
    // peers — their last consumers, the previous update and its displacement pass, are enqueued)
    wait_arrived(kXRows, slot, 0, world, st, chbit(kXGrads, slot) | chbit(kXIds, slot ^ 1));
    ShardGatherArgs A{};
    A.st = ConstStatics(ms.d_st);
    A.in = snd_rows;
    A.out = out;
    A.slot_off = slot_off[slot];
    A.slot = uint32_t(slot);
    A.n_max = uint32_t(max_batch);
    gather_tabs(slot);
    for (uint32_t t0 = 0; t0 < T; t0 += uint32_t(kMaxStepTables)) {
      const uint32_t tc = chunk_tables(t0);
      A.t0 = t0;
      A.tc = tc;
      fill_tabs(A.tab, t0, tc);
      uint32_t gx = 0;
      bool w4 = false, w1 = false;
      for (uint32_t i = 0; i < tc; ++i) {
        A.gt[i] = gt_all[t0 + i];
        gx = std::max(gx, A.gt[i].nblk_items + A.gt[i].nblk_ids);
        if (A.gt[i].n) ((A.gt[i].gv & 1u) ? w1 : w4) = true;
      }
      if (!gx) continue;
      if (w4) LAUNCH_HOT(kTagShardGather, shard_scatter_kernel<4>, dim3(gx, tc), 256, st, A);
      if (w1) LAUNCH_HOT(kTagShardGather, shard_scatter_kernel<1>, dim3(gx, tc), 256, st, A);
      launches += (w4 ? 1u : 0u) + (w1 ? 1u : 0u);
      HIP_OK(hipGetLastError());
    }
  }

  // the received blocks' counts, a step (or more) late: see est_n
  void poll_counts() {
    if (!rcnt_pending || hipEventQuery(ev_rcnt) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    rcnt_pending = false;
    for (uint32_t t = 0; t < T; ++t) {
      int64_t m = 0;
      for (int p = 0; p < world; ++p) m = std::max(m, h_rcnt[size_t(p) * hdr_words + t]);
      m = std::min<int64_t>(std::max<int64_t>(m, 0), int64_t(cap));
      est_n[t] = uint32_t(std::min<int64_t>(int64_t(cap), m + m / 4 + 64));   // (+25 %: the next batch is another one)
    }
  }
  void fetch_counts_lazily(int slot, hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (rcnt_pending || (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)) return;
    const size_t w = size_t(hdr_words) * sizeof(int64_t);
    HIP_OK(hipMemcpy2DAsync(h_rcnt, w, ids_recv[slot], size_t(geo.ids_block) * 8, w, size_t(world),
                            hipMemcpyDeviceToHost, st));
    HIP_OK(hipEventRecord(ev_rcnt, st));
    rcnt_pending = true;
  }
  // ids a (peer, table) block of table t is sized for in the owner's launches
  uint32_t sized_n(uint32_t t) const {
    static const bool by_cap = getenv("MHTE_SHARD_SIZE_BY_CAP") != nullptr;   // (A/B: round 4's sizing)
    return (by_cap || est_n[t] == 0) ? cap : std::min(cap, est_n[t]);
  }

  static void flush_slow_cb(void* ctx, hipStream_t st) { static_cast<ShardStep*>(ctx)->flush_slow(st); }
  void set_hooks(bool on) {
    for (uint32_t t = 0; t < T; ++t) {
      mt->tables[t]->ext_flush = on ? &ShardStep::flush_slow_cb : nullptr;
      mt->tables[t]->ext_ctx = on ? this : nullptr;
    }
  }
  // the owed displacement pass as a launch of its own (anything but the next owner lookup came first)
  void flush_slow(hipStream_t st) {
    if (!slow_pending.exchange(false)) return;
    set_hooks(false);
    if (slow_ev_valid && st != slow_stream) HIP_OK(hipStreamWaitEvent(st, slow_ev, 0));
    slow_ev_valid = false;
    for (OwnerChunk& k : apply_chunks) {
      k.A.zero_headers = 1u;
      shard_slow_all_kernel<<<k.tc, 64, 0, st>>>(k.A);
      HIP_OK(hipGetLastError());
      ++launches;
    }
    if (slow_slot >= 0) hdr_dirty[slow_slot] = false;
  }
  // the pass may ride in a lookup launch when every table's rows take the BASIC update code there
  // ... and with ONE rank only: with more, the pass needs the cross-peer slots of its deferred ids (oslot /
  // xs), which the lookup of the same launch is already rewriting for the next batch
  bool can_fold() const {
    if (world > 1) return false;
    // (a launch of its own is ~6 us beside one table's ~50 us step and ~15 us — a wavefront per table, each
    // with its round trips — beside 26 tables' 450: measured 452 us folded against 465)
    static const uint32_t max_t = getenv("MHTE_SHARD_FOLD_MAX_TABLES") ? uint32_t(atoi(getenv("MHTE_SHARD_FOLD_MAX_TABLES"))) : (1u << 30);
    if (!fold_slow || T > max_t) return false;
    for (uint32_t t = 0; t < T; ++t)
      if (!mt->tables[t]->basic_opts() || mt->tables[t]->has_group_opt) return false;
    return true;
  }

  void clear_x(hipStream_t st) {
    if (!xs) return;
    const uint64_t nslots = uint64_t(T) * (uint64_t(xcap) + 1);
    shard_x_clear_kernel<<<uint32_t(std::min<uint64_t>((nslots + 255) / 256, 4096)), 256, 0, st>>>(xs, nslots, xstride);
    HIP_OK(hipGetLastError());
    x_dirty = false;
  }
  // a table with an occurrence filter keeps the per-peer form: its filter's window moves between senders
  bool per_peer_owner() const {
    if (legacy_owner) return true;
    for (uint32_t t = 0; t < T; ++t)
      if (mt->tables[t]->flt_slots) return true;
    return false;
  }

  // rows back -> occurrences of the batch in `slot` AND the run dedup of (ids_next, split_next) into
  // slot_next, ONE launch (shard_scatter_dedup_kernel; T <= kMaxStepTables).  Tables that move one float
  // per lane are scattered by shard_scatter_kernel<1>.
  bool fuse_scatter = true;     // MHTE_SHARD_FUSE_SCATTER=0: two launches (A/B)
  void scatter_dedup(float* out, int slot, const int64_t* ids_next, const int64_t* split_next, int slot_next,
                     hipStream_t st) {
    wait_arrived(kXRows, slot, 0, world, st, chbit(kXGrads, slot) | chbit(kXIds, slot ^ 1));
    if (ms.stage[slot_next] == 1) ms.clear_slots(1u << slot_next, st);
    for (uint32_t t = 0; t < T; ++t) ms.n_slot[slot_next][t] = uint32_t(split_next[t + 1] - split_next[t]);
    ms.has_hints[slot_next] = false;
    ShardGatherArgs A{};
    A.st = ConstStatics(ms.d_st);
    A.in = snd_rows;
    A.out = out;
    A.slot_off = slot_off[slot];
    A.slot = uint32_t(slot);
    A.n_max = uint32_t(max_batch);
    A.t0 = 0;
    A.tc = T;
    gather_tabs(slot);
    fill_tabs(A.tab, 0, T);
    MDedupArgs D{};
    D.st = ConstStatics(ms.d_st);
    D.ids = ids_next + split_next[0];
    D.slot = uint32_t(slot_next);
    D.T = T;
    MFwdFuse F{};
    uint32_t dblocks = 0, active = 0, gx1 = 0, lin = 0;
    for (uint32_t t = 0; t < T; ++t) {
      A.gt[t] = gt_all[t];
      D.id_off[t] = uint32_t(split_next[t] - split_next[0]);
      D.blk_start[t] = dblocks;
      dblocks += uint32_t((split_next[t + 1] - split_next[t] + kRdBlock - 1) / kRdBlock);
      active += (A.gt[t].n && !(A.gt[t].gv & 1u)) ? 1u : 0u;
    }
    D.id_off[T] = uint32_t(split_next[T] - split_next[0]);
    D.blk_start[T] = dblocks;
    F.nd = std::min<uint32_t>(dblocks, ms.fused_dedup_wgs ? ms.fused_dedup_wgs : uint32_t(ms.num_cus) / 2);
    F.period = 1;
    // two 1024-thread workgroups per CU are resident; the dedup's take their share, the scatter gets the
    // rest ONCE (the distinct ids are dealt out over whatever workgroups a table has: a second dispatch
    // round would only add latency)
    const uint32_t slots = uint32_t(2 * ms.num_cus);
    static const uint32_t ovs_env = getenv("MHTE_SHARD_SCATTER_OVS") ? uint32_t(std::max(1, atoi(getenv("MHTE_SHARD_SCATTER_OVS")))) : 1u;
    const uint32_t room = std::max<uint32_t>(slots > F.nd ? slots - F.nd : 8u, 8u) * ovs_env;
    for (uint32_t t = 0; t < T; ++t) {
      F.fwd_start[t] = lin;
      const ShardGatherTab& g = A.gt[t];
      if (!g.n) continue;
      if (g.gv & 1u) {
        gx1 = std::max(gx1, g.nblk_items + g.nblk_ids);
        continue;
      }
      const uint32_t ids_per_wg = uint32_t(kRdBlock) / shape_lanes(g.gv) * uint32_t(MHTE_SHARD_SCATTER_UNR);
      const uint32_t one_trip = (g.n + ids_per_wg - 1) / ids_per_wg;
      const uint32_t share = std::max<uint32_t>(2, room / std::max(1u, active));
      lin += std::max<uint32_t>(1, std::min(one_trip, share));
    }
    F.fwd_start[T] = lin;
    if (gx1) {
      LAUNCH_HOT(kTagShardGather, shard_scatter_kernel<1>, dim3(gx1, T), 256, st, A);
      ++launches;
    }
    if (F.nd + lin) {
      // (a heavy work item — ~256 occurrences of one id — is shared by four wavefronts: one wavefront walks
      // it 64 positions at a time with a dependent position load per pass, and there are idle ones)
      static const uint32_t split_env = getenv("MHTE_SHARD_ITEM_SPLIT") ? uint32_t(std::max(1, atoi(getenv("MHTE_SHARD_ITEM_SPLIT")))) : 4u;
      LAUNCH_HOT(kTagShardGather, shard_scatter_dedup_kernel, F.nd + lin, kRdBlock, st, A, D, F,
                 std::max<uint32_t>(split_env, ms.item_target / kItemTarget));
      ++launches;
    }
    HIP_OK(hipGetLastError());
    ms.stage[slot_next] = 1;
    disp[slot_next] = false;
    ids_exchanged[slot_next] = false;
    ids_hdr_sent[slot_next] = false;
  }

  // arguments of the owner-side launches for tables [t0, t0 + tc)
  void owner_args(ShardOwnerArgs& A, int slot, bool apply, uint32_t t0, uint32_t tc) const {
    A.views = ConstViews(mt->d_views.p);
    A.geo = geo;
    A.x.orec = orec;
    A.x.oslot = oslot;
    A.x.xs = xs;
    A.x.xmask = xcap ? xcap - 1u : 0u;
    A.x.xstride = xstride;
    A.x.hints = 0;
    A.recv_ids = ids_recv[slot];
    A.rows = apply ? apply_grads() : own_rows;
    A.flags = d_flags;
    A.t0 = t0;
    A.tc = tc;
    fill_tabs(A.tab, t0, tc);
    static const bool no_fast = getenv("MHTE_SHARD_NO_FAST_APPLY") != nullptr;   // (A/B)
    for (uint32_t i = 0; i < tc; ++i) {
      const Table& tb = *mt->tables[t0 + i];
      A.g[i] = uint8_t(seg_shape_code(tb));
      A.count_hits[i] = tb.count_hits ? 1 : 0;
      A.fast[i] = (!no_fast && tb.basic_opts() && !tb.has_group_opt) ? 1 : 0;
    }
  }

  void owner_lookup(int slot, hipStream_t st) {
    wait_arrived(kXIds, slot, 0, world, st, chbit(kXRows, slot));
    if (direct) require_credits(chbit(kXRows, slot), st);   // (the rows go straight into the peers' windows)
    const bool per_peer = per_peer_owner();
    if (xs && !per_peer) {
      if (x_dirty) clear_x(st);   // (a batch that was looked up and never trained left its ids registered)
      x_dirty = true;
    }
    own_slot = slot;
    poll_counts();
    for (uint32_t t = 0; t < T; ++t) own_epoch[t] = mt->tables[t]->mut_epoch;
    const bool fold = !per_peer && slow_pending.exchange(false);   // (claimed: a hook firing now finds nothing)
    if (!fold) flush_slow(st);
    for (uint32_t t0 = 0; t0 < T; t0 += uint32_t(kMaxStepTables)) {
      const uint32_t tc = chunk_tables(t0);
      ShardOwnerArgs A{};
      owner_args(A, slot, false, t0, tc);
      if (per_peer) A.x.xs = nullptr, A.x.oslot = nullptr;   // (no registration: nobody would consume it)
      if (direct) {
        A.peer_win = d_peer_win;
        A.peer_rows_off = win_off_rows + size_t(rank) * size_t(geo.rows_block) * sizeof(float);
      }
      if (fold) {   // the previous update's displacement pass: that update's arguments for these tables
        const ShardOwnerArgs& P = apply_chunks[t0 / uint32_t(kMaxStepTables)].A;
        A.slow_on = 1u;
        A.slow_ids = P.recv_ids;
        A.slow_rows = P.rows;
        A.zero_headers = 1u;
        A.clear_ids = P.clear_ids;
        for (uint32_t i = 0; i < tc; ++i) {
          A.pending[i] = P.pending[i];
          A.a[i] = P.a[i];
        }
      }
      const uint32_t unr = 2u;   // ids per lane group in flight (4: measured no faster on 26 tables, profiles/r05)
      uint32_t gx = 1;
      for (uint32_t i = 0; i < tc; ++i)
        gx = std::max(gx, uint32_t((uint64_t((sized_n(t0 + i) + unr - 1) / unr) * shape_lanes(A.g[i]) + 511) / 512));
      // (grid-stride inside: enough workgroups to fill the chip a few times over, not one per slot)
      const uint32_t fill = std::max<uint32_t>(8, uint32_t(ms.num_cus) * 16 / (uint32_t(world) * tc));
      gx = std::min(gx, fill);
      bool w4 = fold, w1 = false;   // (the float4 instance runs the owed pass)
      for (uint32_t i = 0; i < tc; ++i) ((A.g[i] & 1u) ? w1 : w4) = true;
      if (fold) gx = std::max(gx, tc);
      const dim3 grid(gx, uint32_t(world) * tc + (fold ? 1u : 0u));
#define MHTE_LOOKUP_LAUNCH(W_)                                                                      \
  do {                                                                                              \
    if (fold) LAUNCH_HOT(kTagShardLookup, (shard_lookup_kernel<W_, true, 2>), grid, 512, st, A);    \
    else LAUNCH_HOT(kTagShardLookup, (shard_lookup_kernel<W_, false, 2>), grid, 512, st, A);        \
  } while (0)
      if (w4) MHTE_LOOKUP_LAUNCH(4);
      if (w1) MHTE_LOOKUP_LAUNCH(1);
#undef MHTE_LOOKUP_LAUNCH
      launches += (w4 ? 1u : 0u) + (w1 ? 1u : 0u);
      HIP_OK(hipGetLastError());
    }
    if (fold) {
      set_hooks(false);
      slow_ev_valid = false;
      if (slow_slot >= 0) hdr_dirty[slow_slot] = false;
    }
    if (direct) ++seq_sent[kChRows];
    fetch_counts_lazily(slot, st);
  }

  void owner_apply(int slot, const float* lrs, int64_t update_time, int64_t global_step, hipStream_t st) {
    for (uint32_t t = 0; t < T; ++t) {
      Table& tb = *mt->tables[t];
      tb.note_update_time(update_time);
      tb.ensure_capacity(uint64_t(cap) * uint64_t(world), st);
      tb.pending.reserve(2 * size_t(cap) * size_t(world) + 2);
    }
    sync_views(mt, st);
    // (direct mode: ONE sync point for the gradients and — a step ahead — the next batch's ids; the row
    // buffer is free for the peers' next lookups: this step's scatter is enqueued)
    if (direct) require_arrived(chbit(kXGrads, slot) | chbit(kXIds, slot) | chbit(kXIds, slot ^ 1), chbit(kXRows, slot), st);
    wait_arrived(kXIds, slot, 0, world, st);
    poll_counts();
    // one set of arguments per kMaxStepTables tables; the same for every peer but `peer` / `zero_headers`
    std::vector<OwnerChunk>& chunks = apply_chunks;
    chunks.resize((T + uint32_t(kMaxStepTables) - 1) / uint32_t(kMaxStepTables));
    int64_t lr_off = 0;
    for (size_t c = 0; c < chunks.size(); ++c) {
      OwnerChunk& k = chunks[c];
      const uint32_t t0 = uint32_t(c) * uint32_t(kMaxStepTables);
      k = OwnerChunk{};
      k.tc = chunk_tables(t0);
      ShardOwnerArgs& A = k.A;
      owner_args(A, slot, true, t0, k.tc);
      uint32_t gx = 1;
      for (uint32_t i = 0; i < k.tc; ++i) {
        Table& tb = *mt->tables[t0 + i];
        A.pending[i] = tb.pending.p;
        ApplyArgs& a = A.a[i];
        for (int j = 0; j < kMaxSegments; ++j) a.lr[j] = (j < int(tb.nseg)) ? lrs[lr_off + j] : 0.f;
        lr_off += tb.nseg;
        a.ts = static_cast<uint32_t>(update_time);
        a.sum_dups = 0;
        a.filter_mode = tb.flt_slots ? 1 : 0;   // an owner asks its filter about every id it does not hold
        a.global_step = global_step;
        gx = std::max(gx, (sized_n(t0 + i) + 256u / shape_lanes(A.g[i]) - 1) / (256u / shape_lanes(A.g[i])));
        k.inst[A.g[i] & 1u][(A.g[i] >> 1) & 1u] = true;
        k.inst3[A.g[i] & 1u][(A.g[i] >> 1) & 1u][A.fast[i] ? 1 : 0] = true;
      }
      const uint32_t fill = std::max<uint32_t>(8, uint32_t(ms.num_cus) * 16 / k.tc);
      k.gx_fill = fill;
      k.gx = std::min(gx, fill);
      A.clear_ids = ids_send[slot];
    }
    if (!per_peer_owner()) {
      // ONE launch for every peer's block (+ the displacement pass): the groups of an id's lowest sender
      // apply its entries in rank order (shard_apply_kernel)
      wait_arrived(kXGrads, slot, 0, world, st);
      if (grad_bits == 16) {
        cvt<false>(ids_recv[slot], own_grads16, apply_grads(), 0, world, st);
        ++launches;
      }
      for (size_t c = 0; c < chunks.size(); ++c) {
        OwnerChunk& k = chunks[c];
        ShardOwnerArgs& A = k.A;
        const uint32_t t0 = uint32_t(c) * uint32_t(kMaxStepTables);
        A.peer = 0;
        A.zero_headers = 1u;
        for (uint32_t i = 0; i < k.tc; ++i)
          if (own_slot == slot && own_epoch[t0 + i] == mt->tables[t0 + i]->mut_epoch) A.x.hints |= 1u << i;
        const uint32_t gx = std::max<uint32_t>(1u, std::min<uint32_t>(k.gx, std::max<uint32_t>(8, k.gx_fill / uint32_t(world))));
        const dim3 grid(gx, uint32_t(world) * k.tc);
#define MHTE_APPLY_LAUNCH(W_, G_, F_)                                                                          \
  do {                                                                                                       \
    if (k.inst3[W_ == 1][G_][F_]) {                                                                          \
      if (world > 1) LAUNCH_HOT(kTagShardUpsert, (shard_apply_kernel<W_, G_ != 0, true, F_ != 0>), grid, 256, st, A);  \
      else LAUNCH_HOT(kTagShardUpsert, (shard_apply_kernel<W_, G_ != 0, false, F_ != 0>), grid, 256, st, A);           \
      ++launches;                                                                                            \
    }                                                                                                        \
  } while (0)
        MHTE_APPLY_LAUNCH(4, 0, 1);
        MHTE_APPLY_LAUNCH(1, 0, 1);
        MHTE_APPLY_LAUNCH(4, 0, 0);
        MHTE_APPLY_LAUNCH(1, 0, 0);
        MHTE_APPLY_LAUNCH(4, 1, 0);
        MHTE_APPLY_LAUNCH(1, 1, 0);
#undef MHTE_APPLY_LAUNCH
        HIP_OK(hipGetLastError());
      }
      x_dirty = false;
      own_slot = -1;
      slow_pending = true;
      slow_slot = slot;
      if (can_fold()) {   // the pass rides in the next owner lookup
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        slow_ev_valid = false;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
          if (!slow_ev) HIP_OK(hipEventCreateWithFlags(&slow_ev, hipEventDisableTiming));
          HIP_OK(hipEventRecord(slow_ev, st));
          slow_stream = st;
          slow_ev_valid = true;
        }
        set_hooks(true);
      } else {
        flush_slow(st);
      }
    } else
    for (int p = 0; p < world; ++p) {
      wait_arrived(kXGrads, slot, p, p + 1, st);   // (a peer's block is applied as soon as it has landed)
      if (grad_bits == 16) {
        cvt<false>(ids_recv[slot], own_grads16, apply_grads(), p, 1, st);
        ++launches;
      }
      for (OwnerChunk& k : chunks) {
        ShardOwnerArgs& A = k.A;
        const uint32_t gx = k.gx, tc = k.tc;
        A.peer = uint32_t(p);
        A.zero_headers = p == world - 1 ? 1u : 0u;
        if (k.inst[0][0]) LAUNCH_HOT(kTagShardUpsert, (shard_upsert_kernel<4, false>), dim3(gx, tc), 256, st, A);
        if (k.inst[1][0]) LAUNCH_HOT(kTagShardUpsert, (shard_upsert_kernel<1, false>), dim3(gx, tc), 256, st, A);
        if (k.inst[0][1]) LAUNCH_HOT(kTagShardUpsert, (shard_upsert_kernel<4, true>), dim3(gx, tc), 256, st, A);
        if (k.inst[1][1]) LAUNCH_HOT(kTagShardUpsert, (shard_upsert_kernel<1, true>), dim3(gx, tc), 256, st, A);
        shard_slow_kernel<<<tc, 64, 0, st>>>(A);
        HIP_OK(hipGetLastError());
        launches += uint32_t(k.inst[0][0]) + uint32_t(k.inst[1][0]) + uint32_t(k.inst[0][1]) + uint32_t(k.inst[1][1]) + 1u;
      }
      for (uint32_t t = 0; t < T; ++t)   // the filter's window moves between senders (one filter for all tables)
        if (mt->tables[t]->flt_slots) {
          mt->tables[t]->filter_maintain(st);
          break;
        }
    }
    if (!slow_pending) hdr_dirty[slot] = false;   // (an owed pass clears the headers when it runs)
    for (uint32_t t = 0; t < T; ++t) {
      ++mt->tables[t]->mut_epoch;
      // (the eviction cadence is checked with the pass still owed: a scan that is due runs it first)
      mt->tables[t]->maybe_evict(st);
    }
  }

  const void* x_src(int kind, int slot) const {
    return kind == kXIds ? static_cast<const void*>(ids_send[slot])
                         : kind == kXRows ? static_cast<const void*>(own_rows)
                                          : (grad_bits == 16 ? static_cast<const void*>(snd_grads16) : snd_grads);
  }
  void* x_dst(int kind, int slot) const {
    return kind == kXIds ? static_cast<void*>(ids_recv[slot])
                         : kind == kXRows ? static_cast<void*>(snd_rows)
                                          : (grad_bits == 16 ? static_cast<void*>(own_grads16) : own_grads);
  }
  size_t x_block(int kind) const {
    return kind == kXIds ? size_t(geo.ids_block) * sizeof(int64_t)
                         : size_t(geo.rows_block) * x_elem(kind);
  }
  size_t x_elem(int kind) const {   // bytes per element of a row block on the wire
    return (kind == kXGrads && grad_bits == 16) ? 2 : sizeof(float);
  }

  // after the id exchange of `slot`: its headers (ids sent to / received from every peer, per table)
  // on their way to the host
  void fetch_counts(int slot, hipStream_t st) {
    if (!exact) return;
    const size_t w = size_t(hdr_words) * sizeof(int64_t);
    HIP_OK(hipMemcpy2DAsync(h_cnt[slot], w, ids_send[slot], size_t(geo.ids_block) * 8, w, size_t(world),
                            hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpy2DAsync(h_cnt[slot] + size_t(world) * hdr_words, w, ids_recv[slot],
                            size_t(geo.ids_block) * 8, w, size_t(world), hipMemcpyDeviceToHost, st));
    HIP_OK(hipEventRecord(ev_cnt[slot], st));
  }
  // rows of table t in the block exchanged with peer p: `sent` = ids this rank sent to p (its rows
  // come back / its gradient sums leave), else ids p sent here
  uint32_t seg_rows(int slot, bool sent, int p, uint32_t t) const {
    const int64_t c = h_cnt[slot][(sent ? 0 : size_t(world) * hdr_words) + size_t(p) * hdr_words + t];
    return uint32_t(std::min<int64_t>(std::max<int64_t>(c, 0), int64_t(cap)));
  }

  // ---- the exact form's pieces ---------------------------------------------------------------------------
  // the slot's counts on the host (its headers were fetched behind their exchange): not a wait when the batch
  // was prepared ahead
  void counts_ready(int slot) {
    if (hipEventQuery(ev_cnt[slot]) != hipSuccess) {
      ++host_waits;
      HIP_OK(hipEventSynchronize(ev_cnt[slot]));
    }
  }
  // bytes of the packed stream between this rank and peer p: `sent` = sized by the ids this rank sent p
  size_t packed_bytes(int kind, int slot, bool sent, int p) const {
    size_t n = 0;
    for (uint32_t t = 0; t < T; ++t)
      n += size_t(shard_packed_bytes(seg_rows(slot, sent, p, t), tab[t].dim, kind == kXIds,
                                     kind == kXGrads && grad_bits == 16));
    return n;
  }
  template <bool UNPACK>
  void pack(int kind, const void* src, void* dst, const int64_t* counts, hipStream_t st) {
    ShardPackArgs A{};
    A.src = static_cast<const char*>(src);
    A.dst = static_cast<char*>(dst);
    A.counts = counts;
    A.geo = geo;
    A.ids = kind == kXIds ? 1u : 0u;
    A.half = (kind == kXGrads && grad_bits == 16) ? 1u : 0u;
    A.tab = d_tab;
    const uint32_t gx = std::max<uint32_t>(4, uint32_t(ms.num_cus) * 4 / uint32_t(world));
    shard_pack_kernel<UNPACK><<<dim3(gx, uint32_t(world)), 256, 0, st>>>(A);
    HIP_OK(hipGetLastError());
    ++launches;
  }
  // which id buffer's headers size what this rank SENDS in an exchange of `kind` (the other one sizes what it
  // receives): gradient sums and ids go out for the ids this rank sent, rows for the ids it was sent
  const int64_t* counts_out(int kind, int slot) const { return kind == kXRows ? ids_recv[slot] : ids_send[slot]; }
  const int64_t* counts_in(int kind, int slot) const { return kind == kXRows ? ids_send[slot] : ids_recv[slot]; }
  void note_pairs(uint64_t pairs) {
    wire_pairs += pairs;
    ++wire_exchanges;
    wire_pairs_max = std::max(wire_pairs_max, pairs);
  }
  // the HEADERS of the slot's id blocks inside an open send / recv group
  void group_id_headers(Rccl& R, int slot, hipStream_t st) {
    const size_t b = x_block(kXIds), hb = size_t(hdr_words) * sizeof(int64_t);
    const char* src = static_cast<const char*>(x_src(kXIds, slot));
    char* dst = static_cast<char*>(x_dst(kXIds, slot));
    for (int p = 0; p < world; ++p) {
      R.ok(R.Send(src + size_t(p) * b, hb, ncclInt8, p, comm, st), "Send");
      R.ok(R.Recv(dst + size_t(p) * b, hb, ncclInt8, p, comm, st), "Recv");
    }
  }
  // the packed payload of one exchange: pack, ONE pair per peer, unpack
  void exchange_packed(int kind, int slot, hipStream_t st, int hdr_slot = -1) {
    Rccl& R = Rccl::get();
    counts_ready(slot);
    const bool out_sent = kind != kXRows;
    pack<false>(kind, x_src(kind, slot), stage_snd, counts_out(kind, slot), st);
    const size_t b = x_block(kind);
    uint64_t pairs = 0;
    R.ok(R.GroupStart(), "GroupStart");
    for (int p = 0; p < world; ++p) {
      const size_t ns = packed_bytes(kind, slot, out_sent, p), nr = packed_bytes(kind, slot, !out_sent, p);
      if (ns) R.ok(R.Send(stage_snd + size_t(p) * b, ns, ncclInt8, p, comm, st), "Send");
      if (nr) R.ok(R.Recv(stage_rcv + size_t(p) * b, nr, ncclInt8, p, comm, st), "Recv");
      pairs += (ns || nr) ? 1u : 0u;
    }
    if (hdr_slot >= 0) {   // the next batch's id headers ride in this group
      group_id_headers(R, hdr_slot, st);
      pairs += uint64_t(world);
    }
    R.ok(R.GroupEnd(), "GroupEnd");
    ++launches;
    note_pairs(pairs);
    pack<true>(kind, stage_rcv, x_dst(kind, slot), counts_in(kind, slot), st);
    if (hdr_slot >= 0) {
      fetch_counts(hdr_slot, st);
      ids_hdr_sent[hdr_slot] = true;
    }
  }

  // block p of the source goes to peer p, block p of the destination comes from peer p
  // hdr_slot (exact form, gradient exchange only): the id HEADERS of that slot's batch cross in the same group
  void exchange_rccl(int kind, int slot, hipStream_t st, int hdr_slot = -1) {
    Rccl& R = Rccl::get();
    const char* src = static_cast<const char*>(x_src(kind, slot));
    char* dst = static_cast<char*>(x_dst(kind, slot));
    const size_t b = x_block(kind);
    if (exact && kind != kXIds) {
      exchange_packed(kind, slot, st, hdr_slot);
      return;
    }
    if (exact && kind == kXIds) {
      // the id blocks exact-size too (whole-batch blocks are world x T x batch ids per rank: ~110 MB
      // each way at 26 tables x 65 536 ids x 8 ranks): the headers cross first (with the previous
      // batch's gradient exchange when this batch was prepared ahead), their counts come to the host,
      // then the occupied part of every segment, packed: one pair per peer
      if (!ids_hdr_sent[slot]) {
        R.ok(R.GroupStart(), "GroupStart");
        group_id_headers(R, slot, st);
        R.ok(R.GroupEnd(), "GroupEnd");
        ++launches;
        note_pairs(uint64_t(world));
        fetch_counts(slot, st);
      }
      ids_hdr_sent[slot] = false;
      exchange_packed(kXIds, slot, st);
      return;
    }
    ++launches;   // (one send / recv group)
    R.ok(R.GroupStart(), "GroupStart");
    for (int p = 0; p < world; ++p) {
      R.ok(R.Send(src + size_t(p) * b, b, ncclInt8, p, comm, st), "Send");
      R.ok(R.Recv(dst + size_t(p) * b, b, ncclInt8, p, comm, st), "Recv");
    }
    R.ok(R.GroupEnd(), "GroupEnd");
    if (kind == kXIds) fetch_counts(slot, st);
  }
};

// the ranks of one process (all of them: a test, or one process driving several tables on one GPU): device
// copies stand in for the links, everything else — the packing, the header / payload split of the exact form,
// the counts on the host — is the code the RCCL transport runs
static void group_id_headers(ShardStep** S, int n, int slot, hipStream_t st) {
  const size_t hb = size_t(S[0]->hdr_words) * sizeof(int64_t);
  for (int r = 0; r < n; ++r)
    for (int p = 0; p < n; ++p) {
      const size_t b = S[r]->x_block(kXIds);
      HIP_OK(hipMemcpyAsync(static_cast<char*>(S[p]->x_dst(kXIds, slot)) + size_t(r) * b,
                            static_cast<const char*>(S[r]->x_src(kXIds, slot)) + size_t(p) * b, hb,
                            hipMemcpyDeviceToDevice, st));
    }
  for (int r = 0; r < n; ++r) {
    S[r]->fetch_counts(slot, st);
    S[r]->ids_hdr_sent[slot] = true;
    S[r]->note_pairs(uint64_t(n));
  }
}
static void group_packed(ShardStep** S, int n, int kind, int slot, hipStream_t st) {
  for (int r = 0; r < n; ++r) {
    S[r]->counts_ready(slot);
    S[r]->pack<false>(kind, S[r]->x_src(kind, slot), S[r]->stage_snd, S[r]->counts_out(kind, slot), st);
  }
  for (int r = 0; r < n; ++r) {
    uint64_t pairs = 0;
    for (int p = 0; p < n; ++p) {
      const size_t b = S[r]->x_block(kind);
      const size_t nb = S[r]->packed_bytes(kind, slot, kind != kXRows, p);   // what rank r sends to p
      if (nb) HIP_OK(hipMemcpyAsync(S[p]->stage_rcv + size_t(r) * b, S[r]->stage_snd + size_t(p) * b, nb,
                                    hipMemcpyDeviceToDevice, st));
      pairs += nb ? 1u : 0u;
    }
    S[r]->note_pairs(pairs);
  }
  for (int r = 0; r < n; ++r)
    S[r]->pack<true>(kind, S[r]->stage_rcv, S[r]->x_dst(kind, slot), S[r]->counts_in(kind, slot), st);
}
// hdr_slot (exact form, gradient exchange): the id HEADERS of that slot's batch cross with this exchange
static void shard_exchange(ShardStep** S, int n, int kind, int slot, hipStream_t st, int hdr_slot = -1) {
  if (n > 1)   // (device copies stand in for the links: one exchange per rank and kind)
    for (int r = 0; r < n; ++r) ++S[r]->launches;
  if (n == 1) {
    if (S[0]->alias) return;
    if (S[0]->ipc) S[0]->exchange_ipc(kind, slot, st);
    else S[0]->exchange_rccl(kind, slot, st, S[0]->exact ? hdr_slot : -1);
    return;
  }
  if (S[0]->exact) {
    if (kind == kXIds) {
      if (!S[0]->ids_hdr_sent[slot]) group_id_headers(S, n, slot, st);
      for (int r = 0; r < n; ++r) S[r]->ids_hdr_sent[slot] = false;
    }
    group_packed(S, n, kind, slot, st);
    if (kind != kXIds && hdr_slot >= 0) group_id_headers(S, n, hdr_slot, st);
    return;
  }
  for (int r = 0; r < n; ++r)
    for (int p = 0; p < n; ++p) {
      const size_t b = S[r]->x_block(kind);
      char* dst = static_cast<char*>(S[p]->x_dst(kind, slot)) + size_t(r) * b;
      const char* src = static_cast<const char*>(S[r]->x_src(kind, slot)) + size_t(p) * b;
      HIP_OK(hipMemcpyAsync(dst, src, b, hipMemcpyDeviceToDevice, st));
    }
}

static void shard_check_group(ShardStep** S, int n) {
  if (n < 1 || !S) throw Error(MHTE_INVALID_ARGUMENT, "shard step: no steps");
  if (n == 1) {
    if (S[0]->local_group_member())
      throw Error(MHTE_FAILED_PRECONDITION, "shard step: created without a communicator for world > 1: "
                                            "drive all its ranks together (mhte_shard_group_*)");
    return;
  }
  for (int r = 0; r < n; ++r) {
    if (!S[r] || S[r]->world != n || S[r]->rank != r || S[r]->comm || S[r]->device != S[0]->device ||
        S[r]->T != S[0]->T || S[r]->cap != S[0]->cap || S[r]->geo.ids_block != S[0]->geo.ids_block ||
        S[r]->geo.rows_block != S[0]->geo.rows_block || S[r]->ms.cur != S[0]->ms.cur)
      throw Error(MHTE_INVALID_ARGUMENT, "shard group: steps must be ranks 0..n-1 of one world, same "
                                         "device, tables and capacities, driven in lockstep");
  }
}

static void shard_forward(ShardStep** S, int n, const ShardFwd* a, int64_t n_split, int64_t n_split_next,
                          int prefetched, hipStream_t st) {
  shard_check_group(S, n);
  const bool has_next = a[0].id_next != nullptr;
  for (int r = 0; r < n; ++r) {
    ShardStep& s = *S[r];
    MultiStep& ms = s.ms;
    if ((a[r].id_next != nullptr) != has_next)
      throw Error(MHTE_INVALID_ARGUMENT, "shard group: every rank or none passes a next batch");
    ms.check_ragged(a[r].split, n_split, "id");
    if (has_next) ms.check_ragged(a[r].split_next, n_split_next, "id_next");
    if (!a[r].id || !a[r].emb) throw Error(MHTE_INVALID_ARGUMENT, "shard step forward: null argument");
    if (!aligned16(a[r].emb)) throw Error(MHTE_INVALID_ARGUMENT, "shard step: embedding must be 16-byte aligned");
    int64_t need = 0;
    for (uint32_t t = 0; t < s.T; ++t)
      need += (a[r].split[t + 1] - a[r].split[t]) * int64_t(s.tab[t].dim);
    if (need > a[r].emb_len)
      throw Error(MHTE_INVALID_ARGUMENT, "embedding buffer too short: need " + std::to_string(need));
    if (prefetched) {
      const int nxt = ms.cur ^ 1;
      bool same = s.ahead && ms.stage[nxt] >= 1;
      for (uint32_t t = 0; same && t < s.T; ++t)
        same = ms.n_slot[nxt][t] == uint32_t(a[r].split[t + 1] - a[r].split[t]);
      if (!same)
        throw Error(MHTE_FAILED_PRECONDITION,
                    "shard step forward: this batch was not deduplicated ahead by the previous forward");
    }
  }
  for (int r = 0; r < n; ++r) {
    ShardStep& s = *S[r];
    HIP_OK(hipSetDevice(s.device));
    s.launches = 0;
    s.wire_pairs = s.wire_exchanges = s.wire_pairs_max = s.host_waits = 0;
    s.join_aux(st);
    s.prepare(st, /*keep_owed_pass=*/true);
    if (prefetched) {
      s.ms.cur ^= 1;
    } else {
      const int other = s.ms.cur ^ 1;   // a batch prepared ahead is dropped
      if (s.ms.stage[other] == 2) s.ms.stage[other] = 0;
      s.disp[other] = false;
      s.dedup(a[r].id, a[r].split, s.ms.cur, st);
    }
    s.ahead = false;
  }
  const int cur = S[0]->ms.cur;
  if (S[0]->ms.stage[cur] == 1) {   // not numbered by a backward call: number + pack + send now
    for (int r = 0; r < n; ++r) S[r]->build_and_sum(cur, -1, nullptr, st);
    shard_exchange(S, n, kXIds, cur, st);
    for (int r = 0; r < n; ++r) S[r]->ids_exchanged[cur] = true;
  } else if (!S[0]->ids_exchanged[cur]) {   // numbered on the step's own stream, not sent there
    shard_exchange(S, n, kXIds, cur, st);
    for (int r = 0; r < n; ++r) S[r]->ids_exchanged[cur] = true;
  }
  for (int r = 0; r < n; ++r) S[r]->owner_lookup(cur, st);
  shard_exchange(S, n, kXRows, cur, st);
  for (int r = 0; r < n; ++r) {
    ShardStep& s = *S[r];
    const bool on_aux = has_next && s.overlap && n == 1;
    if (has_next && !on_aux && s.fuse_scatter && s.T <= uint32_t(kMaxStepTables)) {
      s.scatter_dedup(a[r].emb, cur, a[r].id_next, a[r].split_next, cur ^ 1, st);
    } else {
      s.scatter(a[r].emb, cur, st);
      if (on_aux) s.prepare_next_on_aux(a[r].id_next, a[r].split_next, cur ^ 1, st);
      else if (has_next) s.dedup(a[r].id_next, a[r].split_next, cur ^ 1, st);
    }
    if (has_next) s.ahead = true;
  }
  for (int r = 0; r < n; ++r) {
    S[r]->flush_signals(st);
    S[r]->launches_fwd = S[r]->launches;
  }
}

static void shard_backward(ShardStep** S, int n, const float* const* grads, const int64_t* grads_len,
                           const float* lrs, int64_t n_lr, int64_t update_time, int64_t global_step,
                           hipStream_t st) {
  shard_check_group(S, n);
  for (int r = 0; r < n; ++r) {
    ShardStep& s = *S[r];
    MultiStep& ms = s.ms;
    if (!s.disp[ms.cur] || ms.stage[ms.cur] != 2)
      throw Error(MHTE_FAILED_PRECONDITION, "shard step backward: no forward batch outstanding");
    if (!grads[r] || !lrs) throw Error(MHTE_INVALID_ARGUMENT, "shard step backward: null argument");
    if (!aligned16(grads[r])) throw Error(MHTE_INVALID_ARGUMENT, "shard step: gradients must be 16-byte aligned");
    int64_t need = 0, need_lr = 0;
    for (uint32_t t = 0; t < s.T; ++t) {
      need += int64_t(ms.n_slot[ms.cur][t]) * int64_t(s.tab[t].dim);
      need_lr += s.mt->tables[t]->nseg;
    }
    if (need > grads_len[r])
      throw Error(MHTE_INVALID_ARGUMENT, "The length of tensor `value` is too short. Currently value" +
                                             std::to_string(grads_len[r]));
    if (need_lr > n_lr)
      throw Error(MHTE_INVALID_ARGUMENT,
                  "The length of tensor `learning_rate` is too short. Currently value" + std::to_string(n_lr));
  }
  const int cur = S[0]->ms.cur;
  const bool build_next = S[0]->ahead && S[0]->ms.stage[cur ^ 1] == 1;
  for (int r = 0; r < n; ++r) {
    HIP_OK(hipSetDevice(S[r]->device));
    S[r]->prepare(st);
    S[r]->build_and_sum(build_next ? (cur ^ 1) : -1, cur, grads[r], st);
    if (S[r]->grad_bits == 16) {   // the sums leave as fp16
      S[r]->cvt<true>(S[r]->ids_send[cur], S[r]->snd_grads, S[r]->snd_grads16, 0, S[r]->world, st);
      ++S[r]->launches;
    }
  }
  // the exact form (RCCL with whole-batch blocks): only the next batch's id HEADERS cross here, in the gradient
  // exchange's group; its packed ids follow at the next forward, when their counts have reached the host — no
  // host wait inside the step (round 5 waited for the counts right here)
  const bool split_ids = build_next && S[0]->exact && !S[0]->alias && !S[0]->ipc;
  shard_exchange(S, n, kXGrads, cur, st, split_ids ? (cur ^ 1) : -1);
  if (build_next && !split_ids) {
    shard_exchange(S, n, kXIds, cur ^ 1, st);
    for (int r = 0; r < n; ++r) S[r]->ids_exchanged[cur ^ 1] = true;
  }
  for (int r = 0; r < n; ++r) {
    S[r]->owner_apply(cur, lrs, update_time, global_step, st);
    S[r]->ms.stage[cur] = 0;
    S[r]->disp[cur] = false;
    S[r]->flush_signals(st);
  }
}

}  // namespace mhte
#endif  // MHTE_SHARD_HOST_H_
