// Host text path: native counterpart of the reference's per-batch Python loop
//   lxrt/entry.py:36-71 convert_sents_to_features  ->  lxrt/tokenization.py:174-348 BasicTokenizer + WordpieceTokenizer.
// The reference spends ~100 us per sentence there on every step (25 ms per 256 questions: more than the whole GPU train
// step).  This file encodes a batch in one call for pure-ASCII sentences - which is where every rule of the reference's
// tokenizer is a byte-level rule: control characters dropped, whitespace split, ASCII lower-casing, the four ASCII
// punctuation ranges (tokenization.py:366-378) as single tokens, never_split specials kept verbatim, greedy
// longest-match WordPiece with "##" continuation pieces, words over 100 characters -> [UNK] (:298-348), truncation to
// max_seq_length - 2, [CLS] ... [SEP], zero padding (entry.py:52-66).  A sentence with any byte >= 0x80 (accents, CJK:
// Unicode normalisation and category tables) is flagged and left to the Python implementation of the same rules.
// Host code only: no device kernels in this translation unit.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>
#include "common.h"
#include "../../include/rgqa.h"

struct rgqa_tokenizer {
    std::unordered_map<std::string, int64_t> vocab;
    int lower = 1;
    int64_t unk = -1, cls = -1, sep = -1;
    int max_piece = 0;          // longest vocabulary entry in bytes (bounds the greedy search)
};

static inline bool tk_is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
// ASCII control characters (category Cc) other than \t \n \r, plus NUL: removed by _clean_text (tokenization.py:276-287)
static inline bool tk_is_dropped(unsigned char c) { return (c < 0x20 && !(c == '\t' || c == '\n' || c == '\r')) || c == 0x7f; }
static inline bool tk_is_punct(unsigned char c) { return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126); }
static inline bool tk_never_split(std::string_view t) { return t == "[UNK]" || t == "[SEP]" || t == "[PAD]" || t == "[CLS]" || t == "[MASK]"; }

extern "C" {

int rgqa_tokenizer_create(const char* vocab_path, int do_lower_case, rgqa_tokenizer** out) {
    RGQA_REQUIRE(vocab_path != nullptr && out != nullptr, "tokenizer_create: null argument");
    FILE* f = fopen(vocab_path, "rb");
    if (f == nullptr) { rgqa_set_error("tokenizer_create: cannot open vocabulary '%s'", vocab_path); return RGQA_ERR_ARG; }
    rgqa_tokenizer* t = new rgqa_tokenizer();
    t->lower = do_lower_case ? 1 : 0;
    std::string line;
    int64_t index = 0;
    int c;
    auto flush = [&]() {
        // load_vocab: vocab[line.strip()] = index, later duplicates overwrite (tokenization.py:48-60)
        size_t b = 0, e = line.size();
        while (b < e && (tk_is_space((unsigned char)line[b]) || line[b] == '\v' || line[b] == '\f')) ++b;
        while (e > b && (tk_is_space((unsigned char)line[e - 1]) || line[e - 1] == '\v' || line[e - 1] == '\f')) --e;
        std::string tok = line.substr(b, e - b);
        if ((int)tok.size() > t->max_piece) t->max_piece = (int)tok.size();
        t->vocab[tok] = index++;
        line.clear();
    };
    bool any = false;
    while ((c = fgetc(f)) != EOF) {
        any = true;
        if (c == '\n') { flush(); any = false; } else line.push_back((char)c);
    }
    if (any) flush();
    fclose(f);
    auto find = [&](const char* k) { auto it = t->vocab.find(k); return it == t->vocab.end() ? (int64_t)-1 : it->second; };
    t->unk = find("[UNK]"); t->cls = find("[CLS]"); t->sep = find("[SEP]");
    if (t->unk < 0 || t->cls < 0 || t->sep < 0) {
        delete t;
        rgqa_set_error("tokenizer_create: vocabulary '%s' lacks [UNK] / [CLS] / [SEP]", vocab_path);
        return RGQA_ERR_ARG;
    }
    *out = t;
    return RGQA_OK;
}

void rgqa_tokenizer_destroy(rgqa_tokenizer* t) { delete t; }

int rgqa_tokenizer_vocab_size(const rgqa_tokenizer* t, int64_t* out) {
    RGQA_REQUIRE(t != nullptr && out != nullptr, "tokenizer_vocab_size: null argument");
    *out = (int64_t)t->vocab.size();
    return RGQA_OK;
}

int rgqa_tokenizer_encode(const rgqa_tokenizer* t, const char* const* sents, int n, int max_seq_length, int64_t* ids, int64_t* mask,
                          int32_t* lengths, uint8_t* needs_python) {
    RGQA_REQUIRE(t != nullptr && sents != nullptr && ids != nullptr && mask != nullptr && lengths != nullptr && needs_python != nullptr,
                 "tokenizer_encode: null argument");
    RGQA_REQUIRE(n >= 0 && max_seq_length >= 2, "tokenizer_encode: n=%d max_seq_length=%d", n, max_seq_length);
    const int T = max_seq_length, room = T - 2;
    std::string word, cand;
    std::vector<int64_t> pieces;
    for (int i = 0; i < n; ++i) {
        int64_t* row = ids + (size_t)i * T;
        int64_t* mrow = mask + (size_t)i * T;
        for (int k = 0; k < T; ++k) { row[k] = 0; mrow[k] = 0; }
        lengths[i] = 0;
        needs_python[i] = 0;
        const unsigned char* s = reinterpret_cast<const unsigned char*>(sents[i]);
        if (s == nullptr) { needs_python[i] = 1; continue; }
        bool ascii = true;
        for (const unsigned char* p = s; *p; ++p) if (*p >= 0x80) { ascii = false; break; }
        if (!ascii) { needs_python[i] = 1; continue; }
        int cnt = 0;                      // wordpieces emitted so far (the reference tokenises everything, then truncates)
        row[0] = t->cls;
        // one wordpiece-level token (already lower-cased / punctuation-split): greedy longest match (tokenization.py:315-346)
        auto emit_word = [&](const std::string& w) {
            if (w.empty()) return;
            if (w.size() > 100) { if (cnt < room) row[1 + cnt] = t->unk; ++cnt; return; }
            pieces.clear();
            size_t start = 0;
            bool bad = false;
            while (start < w.size()) {
                size_t end = w.size();
                const size_t maxlen = (size_t)t->max_piece;
                if (start == 0) { if (end - start > maxlen) end = start + maxlen; }
                else if (end - start + 2 > maxlen) end = start + (maxlen > 2 ? maxlen - 2 : 0);
                int64_t cur = -1;
                while (start < end) {
                    cand.clear();
                    if (start > 0) cand = "##";
                    cand.append(w, start, end - start);
                    auto it = t->vocab.find(cand);
                    if (it != t->vocab.end()) { cur = it->second; break; }
                    --end;
                }
                if (cur < 0) { bad = true; break; }
                pieces.push_back(cur);
                start = end;
            }
            if (bad) { if (cnt < room) row[1 + cnt] = t->unk; ++cnt; return; }
            for (int64_t id : pieces) { if (cnt < room) row[1 + cnt] = id; ++cnt; }
        };
        const unsigned char* p = s;
        while (*p) {
            while (*p && (tk_is_space(*p) || tk_is_dropped(*p))) {
                // dropped characters vanish BEFORE the whitespace split: "a\x01b" is one token "ab" - handled below; here only
                // leading separators of a token are skipped (a dropped byte between separators separates nothing)
                ++p;
            }
            if (!*p) break;
            // collect one whitespace-delimited token with dropped bytes removed
            word.clear();
            while (*p && !tk_is_space(*p)) { if (!tk_is_dropped(*p)) word.push_back((char)*p); ++p; }
            if (word.empty()) continue;
            if (tk_never_split(word)) { emit_word(word); continue; }
            if (t->lower) for (char& ch : word) if (ch >= 'A' && ch <= 'Z') ch = (char)(ch - 'A' + 'a');
            if (tk_never_split(word)) { emit_word(word); continue; }        // cannot happen after lower-casing ("[unk]"), kept for symmetry with :205-209
            // split on punctuation: every punctuation byte is its own token (:229-247)
            std::string part;
            for (char ch : word) {
                if (tk_is_punct((unsigned char)ch)) {
                    emit_word(part); part.clear();
                    emit_word(std::string(1, ch));
                } else part.push_back(ch);
            }
            emit_word(part);
        }
        const int kept = cnt < room ? cnt : room;
        row[1 + kept] = t->sep;
        const int len = kept + 2;
        for (int k = 0; k < len; ++k) mrow[k] = 1;
        lengths[i] = len;
    }
    return RGQA_OK;
}

}  // extern "C"
