// Host-side driver for the checkpoint codec (monolith_amd/csrc/mhte_ckpt.h): the CPU suite runs it
// against golden bytes, the protobuf runtime and an independent Python reader / writer
// (tests/test_ckpt_codec.py).  Commands:
//   crc <string>                         -> crc32c and masked crc, hex
//   crcsoft <string>                     -> crc32c by the sliced tables (no SSE4.2), hex
//   entry <id> <ts> <dim> <nseg> {<kind> <dim>}... <row floats...>  -> EntryDump bytes, hex
//   decode <hexfile> <dim> <nseg> {<kind> <dim>}...                 -> id ts row...
//   write <path> <snappy 0|1> <n> <len>  -> n records of len bytes, record i filled with byte i
//   writeparts <path> <snappy 0|1> <parts> <n> <len>  -> the save path's pattern: <parts> buffers of n
//                                           framed records each (record j of part p filled with byte
//                                           p + j, len + j % 7 bytes) through write_framed
//   read <path> <snappy 0|1>             -> "<count> <xor of all bytes> <sum of lengths>"
//   pipeline <chunks> <failing stage 0-2 | -1> <failing chunk> [<end>]  -> run_pipeline3 over buffers
//                                           that check the hand-over order: "ok <chunks written>" /
//                                           ERROR; <end>: stage A reports the end at that chunk
//   readbatch <path> <snappy 0|1> <stretch bytes> <threads>  -> the same through read_batch (the
//                                           restore path: stretches, blocks unpacked on <threads>)
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../monolith_amd/csrc/mhte_ckpt.h"

using namespace mhte::ckpt;

static std::vector<SegLayout> segs_from(char** a, int nseg, int dim, int* row_floats) {
  std::vector<SegLayout> v;
  int w = 0, st = dim;
  for (int i = 0; i < nseg; ++i) {
    SegLayout s;
    s.kind = atoi(a[2 * i]);
    s.dim = atoi(a[2 * i + 1]);
    s.w_off = w;
    s.st_off = st;
    w += s.dim;
    {
      const DumpSpec ds = dump_spec(s.kind);
      st += ds.nvec * s.dim + (ds.has_slot() ? 4 : 0);
    }
    v.push_back(s);
  }
  *row_floats = st;
  return v;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string cmd = argv[1];
  try {
    if (cmd == "crc") {
      const std::string s = argv[2];
      printf("%08x %08x\n", crc32c(s.data(), s.size()), masked_crc(s.data(), s.size()));
    } else if (cmd == "crcsoft") {  // the table-driven form, whatever the host CPU offers
      const std::string s = argv[2];
      printf("%08x\n", crc32c_soft(0xffffffffu, reinterpret_cast<const uint8_t*>(s.data()), s.size()) ^ 0xffffffffu);
    } else if (cmd == "entry") {
      const long long id = atoll(argv[2]);
      const unsigned ts = unsigned(strtoul(argv[3], nullptr, 10));
      const int dim = atoi(argv[4]), nseg = atoi(argv[5]);
      int rf;
      std::vector<SegLayout> segs = segs_from(argv + 6, nseg, dim, &rf);
      std::vector<float> row(rf);
      for (int i = 0; i < rf; ++i) row[i] = strtof(argv[6 + 2 * nseg + i], nullptr);
      std::string out;
      encode_entry(out, id, row.data(), segs, dim, ts);
      for (unsigned char c : out) printf("%02x", c);
      printf("\n");
    } else if (cmd == "decode") {
      FILE* f = fopen(argv[2], "rb");
      std::string hex;
      char buf[4096];
      size_t n;
      while ((n = fread(buf, 1, sizeof(buf), f)) > 0) hex.append(buf, n);
      fclose(f);
      std::string raw;
      for (size_t i = 0; i + 1 < hex.size(); i += 2) raw.push_back(char(strtol(hex.substr(i, 2).c_str(), nullptr, 16)));
      const int dim = atoi(argv[3]), nseg = atoi(argv[4]);
      int rf;
      std::vector<SegLayout> segs = segs_from(argv + 5, nseg, dim, &rf);
      std::vector<float> row(rf, -7.f);
      int64_t id;
      uint32_t ts;
      decode_entry(reinterpret_cast<const uint8_t*>(raw.data()), raw.size(), segs, dim, &id, row.data(), &ts);
      printf("%lld %u", (long long)id, ts);
      for (float x : row) printf(" %.9g", x);
      printf("\n");
    } else if (cmd == "recode") {
      // recode <in> <out> <batch 0|1> <dim> <nseg> {<kind> <dim>}...: an UNCOMPRESSED TFRecord file of
      // EntryDump (the single-table layout, hash_table_save_op.cc:147-160) read record by record
      // (batch 0) or through read_batch (batch 1: the restore path), every record decoded into a row
      // and encoded again, written through RecordWriter -> prints "<records> <ids xor> <max ts>"
      const bool batch = atoi(argv[4]) != 0;
      const int dim = atoi(argv[5]), nseg = atoi(argv[6]);
      int rf;
      std::vector<SegLayout> segs = segs_from(argv + 7, nseg, dim, &rf);
      RecordReader r(argv[2], false);
      RecordWriter w(argv[3], false);
      unsigned long long cnt = 0, x = 0;
      unsigned max_ts = 0;
      std::string out;
      auto one = [&](const uint8_t* p, size_t n) {
        std::vector<float> row(rf, 0.f);
        int64_t id;
        uint32_t ts;
        decode_entry(p, n, segs, dim, &id, row.data(), &ts);
        encode_entry(out, id, row.data(), segs, dim, ts);
        w.write(out);
        ++cnt;
        x ^= (unsigned long long)id;
        if (ts > max_ts) max_ts = ts;
      };
      if (!batch) {
        std::string rec;
        while (r.read(&rec)) one(reinterpret_cast<const uint8_t*>(rec.data()), rec.size());
      } else {
        ByteArena arena;
        std::vector<RecordReader::RecRef> refs;
        while (r.read_batch(arena, 1024, refs))
          for (const auto& ref : refs) {
            r.verify(arena.data() + ref.off, ref.len, ref.crc);
            one(reinterpret_cast<const uint8_t*>(arena.data() + ref.off), ref.len);
          }
      }
      w.close();
      printf("%llu %llu %u\n", cnt, x, max_ts);
    } else if (cmd == "write") {
      RecordWriter w(argv[2], atoi(argv[3]) != 0);
      const int n = atoi(argv[4]), len = atoi(argv[5]);
      for (int i = 0; i < n; ++i) w.write(std::string(size_t(len + i % 7), char(i)));
      w.close();
    } else if (cmd == "writeparts") {
      RecordWriter w(argv[2], atoi(argv[3]) != 0);
      const int parts = atoi(argv[4]), n = atoi(argv[5]), len = atoi(argv[6]);
      for (int p = 0; p < parts; ++p) {
        std::string part;
        for (int j = 0; j < n; ++j) RecordWriter::frame(part, std::string(size_t(len + j % 7), char(p + j)));
        w.write_framed(part);
      }
      w.close();
    } else if (cmd == "read") {
      RecordReader r(argv[2], atoi(argv[3]) != 0);
      std::string rec;
      unsigned long long cnt = 0, total = 0;
      unsigned x = 0;
      while (r.read(&rec)) {
        ++cnt;
        total += rec.size();
        for (unsigned char c : rec) x ^= c;
      }
      printf("%llu %u %llu\n", cnt, x, total);
    } else if (cmd == "pipeline") {
      const size_t n = size_t(atoll(argv[2]));
      const int fail_stage = atoi(argv[3]);
      const size_t fail_chunk = size_t(atoll(argv[4]));
      const size_t end_at = argc > 5 ? size_t(atoll(argv[5])) : size_t(-1);
      // what a slot holds: chunk + 1 once its producer has filled it, 0 once its consumer is done
      long ab[2] = {0, 0}, bc[2] = {0, 0};
      size_t written = 0;
      unsigned seed = 12345;
      auto jitter = [&seed](int stage) {   // (only stage A draws: one thread)
        seed = seed * 1664525u + 1013904223u;
        if (stage == 0 && (seed >> 28) == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
      };
      auto boom = [&](int stage, size_t c) {
        if (stage == fail_stage && c == fail_chunk) throw std::runtime_error("stage failed as asked");
      };
      run_pipeline3(
          n,
          [&](size_t c, int slot) -> bool {
            boom(0, c);
            jitter(0);
            if (c == end_at) return false;
            if (ab[slot] != 0) throw std::runtime_error("A overwrote a buffer B had not taken");
            ab[slot] = long(c) + 1;
            return true;
          },
          [&](size_t c, int slot) {
            boom(1, c);
            if (ab[slot] != long(c) + 1) throw std::runtime_error("B saw the wrong chunk");
            if (bc[slot] != 0) throw std::runtime_error("B overwrote a buffer C had not taken");
            if (c % 3 == 0) std::this_thread::sleep_for(std::chrono::microseconds(150));
            bc[slot] = long(c) + 1;
            ab[slot] = 0;
          },
          [&](size_t c, int slot) {
            boom(2, c);
            if (bc[slot] != long(c) + 1) throw std::runtime_error("C saw the wrong chunk");
            if (c % 5 == 0) std::this_thread::sleep_for(std::chrono::microseconds(300));
            if (written != c) throw std::runtime_error("C out of order");
            ++written;
            bc[slot] = 0;
          });
      printf("ok %zu\n", written);
    } else if (cmd == "readbatch") {
      RecordReader r(argv[2], atoi(argv[3]) != 0);
      const size_t stretch = size_t(atoll(argv[4]));
      const int threads = atoi(argv[5]);
      ParallelFor par;
      if (threads > 1)
        par = [threads](size_t n, const std::function<void(size_t, size_t)>& fn) {
          std::vector<std::thread> th;
          const size_t parts = std::min<size_t>(size_t(threads), n);
          for (size_t k = 0; k < parts; ++k)
            th.emplace_back([&, k] { fn(n * k / parts, n * (k + 1) / parts); });
          for (auto& t : th) t.join();
        };
      ByteArena arena;
      std::vector<RecordReader::RecRef> refs;
      unsigned long long cnt = 0, total = 0, batches = 0;
      unsigned x = 0;
      while (r.read_batch(arena, stretch, refs, par)) {
        ++batches;
        for (const auto& ref : refs) {
          r.verify(arena.data() + ref.off, ref.len, ref.crc);
          ++cnt;
          total += ref.len;
          for (uint32_t i = 0; i < ref.len; ++i) x ^= (unsigned char)arena.data()[ref.off + i];
        }
      }
      printf("%llu %u %llu %llu\n", cnt, x, total, batches);
    } else {
      return 2;
    }
  } catch (const std::exception& e) {
    printf("ERROR %s\n", e.what());
    return 1;
  }
  return 0;
}
