// AddressSanitizer/UBSan driver for the host-side C++ of libproqa_hip (npy_io.cpp, common.cpp):
// round trips, partial row IO and a set of malformed headers.  Built and run by
// tests/test_native_asan.py with g++ -fsanitize=address,undefined (CPU only; GPU ASAN is not
// available on the pool).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/proqa_hip.h"

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #cond, __LINE__, proqa_last_error()); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static void write_raw(const std::string& path, const std::string& bytes) {
  FILE* f = fopen(path.c_str(), "wb");
  fwrite(bytes.data(), 1, bytes.size(), f);
  fclose(f);
}

static std::string header(const char* dict, int major = 1) {
  std::string h(dict);
  h.push_back('\n');
  std::string out("\x93NUMPY", 6);
  out.push_back((char)major);
  out.push_back(0);
  if (major == 1) {
    out.push_back((char)(h.size() & 0xff));
    out.push_back((char)(h.size() >> 8));
  } else {
    for (int i = 0; i < 4; ++i) out.push_back((char)((h.size() >> (8 * i)) & 0xff));
  }
  return out + h;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const std::string p = dir + "/asan_a.npy";
  std::vector<uint16_t> a(7 * 128);
  for (size_t i = 0; i < a.size(); ++i) a[i] = (uint16_t)(i * 31);
  CHECK(proqa_npy_write(p.c_str(), a.data(), 7, 128, PROQA_F16) == 0);
  proqa_npy_info info;
  CHECK(proqa_npy_stat(p.c_str(), &info) == 0);
  CHECK(info.rows == 7 && info.cols == 128 && info.dtype == PROQA_F16 && info.data_offset % 64 == 0);
  std::vector<uint16_t> b(3 * 128);
  CHECK(proqa_npy_read_rows(p.c_str(), 2, 3, b.data(), b.size() * 2) == 0);
  CHECK(memcmp(b.data(), a.data() + 2 * 128, b.size() * 2) == 0);
  CHECK(proqa_npy_read_rows(p.c_str(), 5, 3, b.data(), b.size() * 2) == PROQA_EINVAL);   // past the end
  CHECK(proqa_npy_read_rows(p.c_str(), 0, 3, b.data(), 10) == PROQA_EINVAL);             // small buffer
  CHECK(proqa_npy_read_rows(p.c_str(), 0, 0, nullptr, 0) == 0);

  const std::string c = dir + "/asan_c.npy";
  CHECK(proqa_npy_create(c.c_str(), 5, 128, PROQA_F32) == 0);
  std::vector<float> f(2 * 128, 1.5f);
  CHECK(proqa_npy_write_rows(c.c_str(), 3, 2, f.data(), 128, PROQA_F32) == 0);
  CHECK(proqa_npy_write_rows(c.c_str(), 4, 2, f.data(), 128, PROQA_F32) == PROQA_EINVAL);   // past the end
  CHECK(proqa_npy_write_rows(c.c_str(), 0, 2, f.data(), 128, PROQA_F16) == PROQA_EINVAL);   // other dtype than the file
  CHECK(proqa_npy_write_rows(c.c_str(), 0, 2, f.data(), 64, PROQA_F32) == PROQA_EINVAL);    // other width
  std::vector<float> g(5 * 128);
  CHECK(proqa_npy_read_rows(c.c_str(), 0, 5, g.data(), g.size() * 4) == 0);
  CHECK(g[0] == 0.f && g[3 * 128] == 1.5f && g[5 * 128 - 1] == 1.5f);
  CHECK(proqa_npy_write(c.c_str(), nullptr, 0, 128, PROQA_F32) == 0);                    // empty array
  CHECK(proqa_npy_stat(c.c_str(), &info) == 0 && info.rows == 0);

  // malformed inputs must fail cleanly
  const std::string m = dir + "/asan_m.npy";
  const char* bad[] = {
      "{'descr': '<i8', 'fortran_order': False, 'shape': (2, 128), }",
      "{'descr': '<f2', 'fortran_order': True, 'shape': (2, 128), }",
      "{'descr': '<f2', 'fortran_order': False, 'shape': (2,), }",
      "{'descr': '<f2', 'fortran_order': False, 'shape': (2, 3, 4), }",
      "{'descr': '<f2', 'fortran_order': False, 'shape': (-2, 128), }",
      "{'descr': '<f2', 'fortran_order': False, 'shape': (2, 128",
      "{'fortran_order': False, 'shape': (2, 128), }",
      "{'descr': '<f2', 'shape': (2, 128), }",
      "{'descr': '<f2', 'fortran_order': False, }",
      "{'descr': '<f2', 'fortran_order': False, 'shape': (99999999999999999999, 128), }",
      "",
  };
  for (const char* dict : bad) {
    write_raw(m, header(dict));
    CHECK(proqa_npy_stat(m.c_str(), &info) != 0);
    CHECK(strlen(proqa_last_error()) > 0);
  }
  write_raw(m, "\x93NUMPY");                     // truncated preamble
  CHECK(proqa_npy_stat(m.c_str(), &info) == PROQA_EFORMAT);
  write_raw(m, std::string("\x93NUMPY\x01\x00\xff\xff", 10) + "{'descr'");   // header length beyond the file
  CHECK(proqa_npy_stat(m.c_str(), &info) == PROQA_EFORMAT);
  write_raw(m, header("{'descr': '<f2', 'fortran_order': False, 'shape': (4, 128), }", 2) + std::string(16, 'x'));
  CHECK(proqa_npy_stat(m.c_str(), &info) == PROQA_EFORMAT);                  // v2 header, data truncated
  write_raw(m, header("{'descr': '<f2', 'fortran_order': False, 'shape': (1, 4), }", 3) + std::string(8, 'x'));
  CHECK(proqa_npy_stat(m.c_str(), &info) == 0 && info.cols == 4);
  CHECK(proqa_npy_stat("/nonexistent/x.npy", &info) == PROQA_EIO);
  CHECK(proqa_npy_stat(nullptr, &info) == PROQA_EINVAL);

  // rand_perm: a permutation, deterministic in the seed
  std::vector<int32_t> perm(1000), perm2(1000);
  CHECK(proqa_rand_perm(1000, 1234, perm.data()) == 0 && proqa_rand_perm(1000, 1234, perm2.data()) == 0);
  CHECK(perm == perm2);
  std::vector<int> seen(1000, 0);
  for (int v : perm) {
    CHECK(v >= 0 && v < 1000);
    seen[v]++;
  }
  for (int s : seen) CHECK(s == 1);
  CHECK(proqa_rand_perm(-1, 1, perm.data()) == PROQA_EINVAL);
  CHECK(proqa_abi_version() == PROQA_ABI_VERSION);

  // native WordPiece: truncation at the row edge, words around the 100-character limit, control bytes, declined texts,
  // several threads, a vocabulary without the special tokens
  {
    const std::string vocab = "[PAD]\n[UNK]\n[CLS]\n[SEP]\nab\n##cd\n##c\n.\nx\n##x";
    proqa_wordpiece* tok = nullptr;
    CHECK(proqa_wordpiece_create(vocab.data(), vocab.size(), 1, &tok) == 0);
    std::vector<std::string> texts = {"", "AB abcd abc. zz", std::string(100, 'x'), std::string(101, 'x'), "ab\x01" "cd \t ab",
                                      "caf\xc3\xa9", "[CLS]", "ab ab ab ab ab ab ab ab ab ab",
                                      // malformed or out-of-table UTF-8 must be handed back, never read past: a cut-off
                                      // three-byte sequence, a lone continuation byte, an overlong form, an emoji, a surrogate
                                      "ab \xe4\xb8", "\x80", "\xc0\xaf", "ab \xf0\x9f\x98\x80", "\xed\xa0\x80",
                                      // table paths: CJK spacing, Hangul decomposition, a stripped mark, Unicode punctuation
                                      "ab\xe4\xb8\xad" "ab", "\xed\x95\x9c", "a\xcc\x81" "b.", "ab\xe2\x80\x94" "ab"};
    std::vector<const char*> ptrs;
    std::vector<int64_t> sizes;
    for (auto& t : texts) {
      ptrs.push_back(t.data());
      sizes.push_back((int64_t)t.size());
    }
    for (int max_length : {2, 3, 8, 128}) {
      for (int threads : {1, 3, 16}) {
        std::vector<int64_t> ids(texts.size() * max_length, -7);
        std::vector<int32_t> lens(texts.size(), -7);
        CHECK(proqa_wordpiece_encode_batch(tok, ptrs.data(), sizes.data(), (int64_t)texts.size(), max_length, ids.data(),
                                           lens.data(), threads) == 0);
        CHECK(lens[0] == 2 && ids[0] == 2 && ids[1] == 3);                       // "" -> [CLS] [SEP]
        CHECK(lens[6] == -1);                                                      // a '[': declined
        // malformed / beyond the BMP: declined (unless the row was full before the bad byte was reached: what follows the
        // truncation point changes no kept token)
        if (max_length >= 8)
          for (int i = 8; i <= 12; ++i) CHECK(lens[i] == -1);
        for (int i = 13; i <= 16; ++i) CHECK(lens[i] >= 2);
        for (size_t i = 0; i < texts.size(); ++i) CHECK(lens[i] == -1 || (lens[i] >= 2 && lens[i] <= max_length));
        if (max_length == 128) {
          CHECK(lens[1] == 9);   // [CLS] ab ab ##cd ab ##c . [UNK] [SEP]
          const int64_t want[9] = {2, 4, 4, 5, 4, 6, 7, 1, 3};
          for (int j = 0; j < 9; ++j) CHECK(ids[1 * 128 + j] == want[j]);
          CHECK(lens[2] == 102 && ids[2 * 128 + 1] == 8 && ids[2 * 128 + 100] == 9);   // 100 x: x ##x ... ##x
          CHECK(lens[3] == 3 && ids[3 * 128 + 1] == 1);                                 // 101 x: [UNK]
          CHECK(lens[4] == 5 && ids[4 * 128 + 1] == 4 && ids[4 * 128 + 2] == 5);        // the control byte vanishes: "abcd"
          CHECK(lens[5] == 3 && ids[5 * 128 + 1] == 1);                                 // "cafe" after the accent is stripped: [UNK]
          CHECK(lens[13] == 5 && ids[13 * 128 + 1] == 4 && ids[13 * 128 + 2] == 1 && ids[13 * 128 + 3] == 4);   // ab [UNK] ab
          CHECK(lens[15] == 4 && ids[15 * 128 + 1] == 4 && ids[15 * 128 + 2] == 7);      // "ab" "." : the acute accent is gone
          CHECK(lens[16] == 5 && ids[16 * 128 + 2] == 1);                                // the em dash stands alone
        }
      }
    }
    {
      // records of a JSON-lines file: plain ones are tokenised, everything cut off or unusual is handed back (-2) without a
      // read past the end of the line
      std::vector<std::string> recs = {"{\"id\": 1, \"text\": \"ab abcd\"}\n", "{\"text\": \"a\\", "{\"text\": \"\\u12", "{\"text\": \"\\u00e9x\"}",
                                       "{\"text\"", "{", "", "{\"text\": \"ab\", \"n\": -", "{\"text\": \"ab\", \"n\": 1e", "{\"text\": \"ab\"} x",
                                       "{\"n\": tru", "  {\"text\":\"ab\",\"k\":null}  "};
      std::vector<const char*> rp;
      std::vector<int64_t> rs;
      for (auto& r : recs) {
        rp.push_back(r.data());
        rs.push_back((int64_t)r.size());
      }
      for (int threads : {1, 5}) {
        std::vector<int64_t> ids(recs.size() * 8, -7);
        std::vector<int32_t> lens(recs.size(), -7);
        CHECK(proqa_wordpiece_encode_jsonl_batch(tok, rp.data(), rs.data(), (int64_t)recs.size(), "text", 8, ids.data(), lens.data(),
                                                 threads) == 0);
        CHECK(lens[0] == 5 && ids[1] == 4 && ids[2] == 4 && ids[3] == 5);   // [CLS] ab ab ##cd [SEP]
        CHECK(lens[3] == 3 && lens[11] == 3);
        for (int i : {1, 2, 4, 5, 6, 7, 8, 9, 10}) CHECK(lens[i] == -2);
      }
      CHECK(proqa_wordpiece_encode_jsonl_batch(tok, rp.data(), rs.data(), 1, nullptr, 8, nullptr, nullptr, 1) == PROQA_EINVAL);
    }
    CHECK(proqa_wordpiece_encode_batch(tok, ptrs.data(), sizes.data(), 1, 1, nullptr, nullptr, 1) == PROQA_EINVAL);
    CHECK(proqa_wordpiece_free(tok) == 0);
    const std::string no_special = "a\nb";
    CHECK(proqa_wordpiece_create(no_special.data(), no_special.size(), 1, &tok) == PROQA_EINVAL);
  }
  printf("asan driver ok\n");
  return 0;
}
