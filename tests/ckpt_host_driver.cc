// Host-side driver for the checkpoint codec (monolith_amd/csrc/mhte_ckpt.h): the CPU suite runs it
// against golden bytes, the protobuf runtime and an independent Python reader / writer
// (tests/test_ckpt_codec.py).  Commands:
//   crc <string>                         -> crc32c and masked crc, hex
//   crcsoft <string>                     -> crc32c by the sliced tables (no SSE4.2), hex
//   entry <id> <ts> <dim> <nseg> {<kind> <dim>}... <row floats...>  -> EntryDump bytes, hex
//   decode <hexfile> <dim> <nseg> {<kind> <dim>}...                 -> id ts row...
//   write <path> <snappy 0|1> <n> <len>  -> n records of len bytes, record i filled with byte i
//   read <path> <snappy 0|1>             -> "<count> <xor of all bytes> <sum of lengths>"
#include <cstdio>
#include <cstdlib>
#include <string>
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
    } else if (cmd == "write") {
      RecordWriter w(argv[2], atoi(argv[3]) != 0);
      const int n = atoi(argv[4]), len = atoi(argv[5]);
      for (int i = 0; i < n; ++i) w.write(std::string(size_t(len + i % 7), char(i)));
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
    } else {
      return 2;
    }
  } catch (const std::exception& e) {
    printf("ERROR %s\n", e.what());
    return 1;
  }
  return 0;
}
