// BERT WordPiece tokenisation of ASCII text, native and multi-threaded (proqa_wordpiece_* in proqa_hip.h).
//
// Replaces `tokenizer.encode(sent, max_length=L)` of /root/reference/retrieval/datasets.py:285-286 (transformers'
// BertTokenizer: BasicTokenizer + WordpieceTokenizer; in current transformers the `tokenizers` library's BertNormalizer /
// BertPreTokenizer / WordPiece / TemplateProcessing) on the host side of get_embed.py, for the texts it can reproduce
// EXACTLY: pure 7-bit ASCII without a '[' (no special-token spelling can occur).  For those the pipeline is
//   clean:     drop NUL and control characters (0x01-0x08, 0x0B, 0x0C, 0x0E-0x1F, 0x7F); \t \n \r -> space
//   lowercase: A-Z -> a-z (uncased models only; accent stripping is a no-op on ASCII)
//   split:     on spaces; every ASCII punctuation character (33-47, 58-64, 91-96, 123-126) is a token of its own
//   wordpiece: a word longer than 100 characters is [UNK]; otherwise greedy longest match, continuation pieces with the
//              "##" prefix; a word with an unmatchable remainder is [UNK] as a whole
//   template:  [CLS] pieces [SEP], truncated on the right to max_length tokens in all
// Every other text (any byte >= 0x80, any '[') is flagged and left to the caller's reference tokenizer: Unicode
// normalisation is not restated here.  tests/test_host_datasets.py compares the two on random ASCII strings.
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"

struct proqa_wordpiece {
  std::string storage;                                   // all tokens, back to back
  std::unordered_map<std::string_view, int32_t> whole;   // tokens that start a word
  std::unordered_map<std::string_view, int32_t> cont;    // continuation pieces, keyed WITHOUT their "##"
  int32_t unk = -1, cls = -1, sep = -1;
  bool lower = true;
  size_t max_whole = 0, max_cont = 0;                    // longest key of either kind (bounds the match loop)
};

namespace {

using namespace proqa;

inline bool ascii_punct(unsigned char c) {
  return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
}

// pieces of one word (already cleaned / lower-cased, no spaces, no punctuation) appended to out; false = the caller stops
// (the row is full)
inline bool emit(std::vector<int32_t>& out, int32_t id, size_t limit) {
  if (out.size() >= limit) return false;
  out.push_back(id);
  return true;
}

bool wordpiece_word(const proqa_wordpiece& t, const char* w, size_t n, std::vector<int32_t>& out, size_t limit,
                    std::vector<int32_t>& scratch) {
  if (n > 100) return emit(out, t.unk, limit);
  scratch.clear();
  size_t start = 0;
  while (start < n) {
    const auto& map = start == 0 ? t.whole : t.cont;
    const size_t longest = start == 0 ? t.max_whole : t.max_cont;
    size_t end = std::min(n, start + longest);
    int32_t id = -1;
    for (; end > start; --end) {
      auto it = map.find(std::string_view(w + start, end - start));
      if (it != map.end()) {
        id = it->second;
        break;
      }
    }
    if (id < 0) return emit(out, t.unk, limit);   // the whole word is unknown
    scratch.push_back(id);
    start = end;
  }
  for (int32_t id : scratch)
    if (!emit(out, id, limit)) return false;
  return true;
}

// one text -> ids_row[0, max_length) (zero-padded), returns its length, or -1 if the text needs the reference tokenizer
int encode_one(const proqa_wordpiece& t, const char* s, size_t n, int max_length, int64_t* ids_row, std::vector<int32_t>& out,
               std::vector<int32_t>& scratch, std::string& word) {
  for (size_t i = 0; i < n; ++i) {
    const unsigned char c = (unsigned char)s[i];
    if (c >= 0x80 || c == '[') return -1;
  }
  const size_t limit = (size_t)std::max(0, max_length - 2);   // pieces kept between [CLS] and [SEP]
  out.clear();
  word.clear();
  bool room = true;
  auto flush = [&]() {
    if (!word.empty() && room) room = wordpiece_word(t, word.data(), word.size(), out, limit, scratch);
    word.clear();
  };
  for (size_t i = 0; i < n && room; ++i) {
    unsigned char c = (unsigned char)s[i];
    if (c == '\t' || c == '\n' || c == '\r') c = ' ';
    if (c < 0x20 || c == 0x7F) continue;                       // NUL and control characters vanish (they do not split a word)
    if (c == ' ') {
      flush();
    } else if (ascii_punct(c)) {
      flush();
      if (room) {
        const char p = (char)c;
        room = wordpiece_word(t, &p, 1, out, limit, scratch);
      }
    } else {
      word.push_back(t.lower && c >= 'A' && c <= 'Z' ? (char)(c + 32) : (char)c);
    }
  }
  flush();
  int len = 0;
  if (max_length >= 1) ids_row[len++] = t.cls;
  for (size_t i = 0; i < out.size() && len < max_length - 1; ++i) ids_row[len++] = out[i];
  if (len < max_length) ids_row[len++] = t.sep;
  for (int i = len; i < max_length; ++i) ids_row[i] = 0;
  return len;
}

}  // namespace

extern "C" {

int proqa_wordpiece_create(const char* vocab, size_t vocab_bytes, int do_lower_case, proqa_wordpiece** out) {
  if (!vocab || !out) return fail(PROQA_EINVAL, "wordpiece_create: NULL argument");
  *out = nullptr;
  proqa_wordpiece* t = new (std::nothrow) proqa_wordpiece();
  if (!t) return fail(PROQA_ENOMEM, "wordpiece_create: out of host memory");
  t->storage.assign(vocab, vocab_bytes);
  t->lower = do_lower_case != 0;
  // tokens separated by '\n', token i has id i (the order of vocab.txt)
  const char* base = t->storage.data();
  size_t pos = 0;
  int32_t id = 0;
  while (pos <= t->storage.size()) {
    size_t nl = t->storage.find('\n', pos);
    if (nl == std::string::npos) nl = t->storage.size();
    if (nl == t->storage.size() && nl == pos) break;   // no trailing empty token
    std::string_view tok(base + pos, nl - pos);
    if (tok.size() > 2 && tok[0] == '#' && tok[1] == '#') {
      t->cont.emplace(tok.substr(2), id);              // (emplace keeps the FIRST id of a duplicated token, like a dict built by update would not -- vocab files have none)
      t->max_cont = std::max(t->max_cont, tok.size() - 2);
    } else {
      t->whole.emplace(tok, id);
      t->max_whole = std::max(t->max_whole, tok.size());
    }
    if (tok == "[UNK]") t->unk = id;
    if (tok == "[CLS]") t->cls = id;
    if (tok == "[SEP]") t->sep = id;
    ++id;
    pos = nl + 1;
  }
  if (t->unk < 0 || t->cls < 0 || t->sep < 0) {
    delete t;
    return fail(PROQA_EINVAL, "wordpiece_create: the vocabulary lacks [UNK], [CLS] or [SEP]");
  }
  *out = t;
  return PROQA_OK;
}

int proqa_wordpiece_free(proqa_wordpiece* t) {
  delete t;
  return PROQA_OK;
}

int proqa_wordpiece_encode_batch(const proqa_wordpiece* t, const char* const* texts, const int64_t* text_bytes, int64_t n,
                                 int max_length, int64_t* ids_out, int32_t* lens_out, int n_threads) {
  if (!t || (n > 0 && (!texts || !text_bytes || !ids_out || !lens_out)) || n < 0 || max_length < 2)
    return fail(PROQA_EINVAL, "wordpiece_encode_batch: bad argument");
  if (n == 0) return PROQA_OK;
  n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n));
  auto work = [&](int64_t lo, int64_t hi) {
    std::vector<int32_t> out, scratch;
    std::string word;
    for (int64_t i = lo; i < hi; ++i)
      lens_out[i] = encode_one(*t, texts[i], (size_t)text_bytes[i], max_length, ids_out + i * max_length, out, scratch, word);
  };
  if (n_threads == 1) {
    work(0, n);
    return PROQA_OK;
  }
  std::vector<std::thread> pool;
  const int64_t per = (n + n_threads - 1) / n_threads;
  for (int th = 0; th < n_threads; ++th) {
    const int64_t lo = th * per, hi = std::min<int64_t>(n, lo + per);
    if (lo < hi) pool.emplace_back(work, lo, hi);
  }
  for (auto& th : pool) th.join();
  return PROQA_OK;
}

}  // extern "C"
