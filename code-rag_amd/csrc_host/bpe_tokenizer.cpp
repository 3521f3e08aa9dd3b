// bpe_tokenizer.cpp -- byte-level BPE tokenizer (GPT-2 / RoBERTa family, the one UniXcoder uses) for the host side of the
// embedding path: text -> token ids, many texts in parallel, straight into a fixed-stride int32 matrix.
//
// Why it exists: the HIP encoder embeds ~4.6 M tokens/s; the Python-facing tokenizers top out near 1 M tokens/s on the same
// box (object creation per encoding), which would cap the plugin surface (EmbeddingProvider.embed_batch) at a fifth of what
// the GPU can take.  Replaces the tokenizer calls of UniXcoder.tokenize (src/lattice/providers/unixcoder_provider.py:105-122:
// RobertaTokenizer.tokenize + convert_tokens_to_ids).  Oracle for parity: the installed HF `tokenizers` / `transformers`
// RobertaTokenizer (tests/test_tokenizer_native.py compares ids on source files and adversarial strings).
//
// Algorithm (restated from the published GPT-2 encoder and the HF ByteLevel / BPE components):
//  1. the text is cut at added (special) tokens -- leftmost match, longest first; a token flagged `lstrip` also swallows
//     the whitespace before it (RoBERTa's <mask>);
//  2. every other stretch is pre-tokenized by the pattern
//       's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+
//     evaluated by hand (alternatives in order, greedy, the one lookahead resolved as "give back the last whitespace when
//     a non-space follows"); the three character classes come from unicode_tables.h, probed from the oracle library;
//  3. the UTF-8 bytes of a piece are mapped to the 256 printable stand-in characters (bytes_to_unicode) and merged pairwise
//     by ascending merge rank until no ranked pair is left; the resulting symbols are looked up in the vocabulary.
// Pieces are cached per thread.  No Python objects are created: the caller gets [n, max_body] int32 and the true lengths.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "unicode_tables.h"

namespace {

enum Cls : uint8_t { OTHER = 0, LETTER = 1, NUMBER = 2, SPACE = 3 };

bool in_ranges(const CpRange *r, int n, unsigned cp)
{
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < r[mid].lo)
            hi = mid - 1;
        else if (cp > r[mid].hi)
            lo = mid + 1;
        else
            return true;
    }
    return false;
}

struct Classifier {
    uint8_t ascii[128];
    Classifier()
    {
        for (unsigned c = 0; c < 128; ++c) ascii[c] = slow(c);
    }
    static uint8_t slow(unsigned cp)
    {
        if (in_ranges(kLetter, kLetterCount, cp)) return LETTER;
        if (in_ranges(kNumber, kNumberCount, cp)) return NUMBER;
        if (in_ranges(kSpace, kSpaceCount, cp)) return SPACE;
        return OTHER;
    }
    uint8_t operator()(unsigned cp) const { return cp < 128 ? ascii[cp] : slow(cp); }
};

// decode one UTF-8 sequence at s[i] (i < n); malformed bytes come back as a single "other" unit of length 1
inline unsigned decode(const unsigned char *s, size_t n, size_t i, int &len)
{
    const unsigned c = s[i];
    if (c < 0x80) {
        len = 1;
        return c;
    }
    const int need = (c >= 0xC2 && c <= 0xDF) ? 1 : (c >= 0xE0 && c <= 0xEF) ? 2 : (c >= 0xF0 && c <= 0xF4) ? 3 : -1;
    if (need < 0 || i + (size_t)need >= n) {
        len = 1;
        return 0xFFFD;
    }
    unsigned cp = need == 1 ? (c & 0x1F) : need == 2 ? (c & 0x0F) : (c & 0x07);
    for (int k = 1; k <= need; ++k) {
        const unsigned cc = s[i + k];
        if ((cc & 0xC0) != 0x80) {
            len = 1;
            return 0xFFFD;
        }
        cp = (cp << 6) | (cc & 0x3F);
    }
    len = need + 1;
    return cp;
}

struct Special {
    std::string text;
    int id;
    bool lstrip;
};

struct Tokenizer {
    std::unordered_map<std::string, int> vocab;          // symbol string (in the byte-stand-in alphabet) -> id
    std::vector<std::string> id_to_sym;
    std::unordered_map<uint64_t, std::pair<int, int>> merges;   // (left id, right id) -> (rank, merged id)
    std::string byte_sym[256];                            // UTF-8 of the stand-in character of each byte
    int byte_id[256];                                     // its vocabulary id (-1: not in the vocabulary)
    std::vector<Special> specials;
    int unk_id = -1;
    Classifier cls;
};

void append_utf8(std::string &out, unsigned cp)
{
    if (cp < 0x80)
        out.push_back((char)cp);
    else if (cp < 0x800) {
        out.push_back((char)(0xC0 | (cp >> 6)));
        out.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        out.push_back((char)(0xE0 | (cp >> 12)));
        out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        out.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

// GPT-2 bytes_to_unicode: printable bytes map to themselves, the rest to 256, 257, ...
void build_byte_alphabet(Tokenizer &t)
{
    bool keep[256] = {false};
    for (int b = '!'; b <= '~'; ++b) keep[b] = true;
    for (int b = 0xA1; b <= 0xAC; ++b) keep[b] = true;
    for (int b = 0xAE; b <= 0xFF; ++b) keep[b] = true;
    int extra = 0;
    for (int b = 0; b < 256; ++b) {
        const unsigned cp = keep[b] ? (unsigned)b : 256u + (unsigned)extra++;
        t.byte_sym[b].clear();
        append_utf8(t.byte_sym[b], cp);
    }
}

struct Worker {
    const Tokenizer &t;
    std::unordered_map<std::string, std::vector<int>> cache;
    std::vector<int> syms;
    explicit Worker(const Tokenizer &tk) : t(tk) { cache.reserve(1 << 14); }

    // BPE of one pre-token (raw UTF-8 bytes of the piece) appended to out
    void bpe(const unsigned char *p, size_t n, std::vector<int> &out)
    {
        std::string key(reinterpret_cast<const char *>(p), n);
        auto it = cache.find(key);
        if (it != cache.end()) {
            out.insert(out.end(), it->second.begin(), it->second.end());
            return;
        }
        syms.clear();
        for (size_t i = 0; i < n; ++i) syms.push_back(t.byte_id[p[i]]);
        bool known = true;
        for (int s : syms) known = known && s >= 0;
        if (known) {
            while (syms.size() > 1) {
                int best_rank = INT32_MAX, best_left = -1, best_right = -1, best_new = -1;
                for (size_t i = 0; i + 1 < syms.size(); ++i) {
                    auto m = t.merges.find(((uint64_t)(uint32_t)syms[i] << 32) | (uint32_t)syms[i + 1]);
                    if (m != t.merges.end() && m->second.first < best_rank) {
                        best_rank = m->second.first;
                        best_left = syms[i];
                        best_right = syms[i + 1];
                        best_new = m->second.second;
                    }
                }
                if (best_left < 0) break;
                size_t w = 0;
                for (size_t i = 0; i < syms.size();) {   // merge every occurrence of the best pair, left to right
                    if (i + 1 < syms.size() && syms[i] == best_left && syms[i + 1] == best_right) {
                        syms[w++] = best_new;
                        i += 2;
                    } else {
                        syms[w++] = syms[i++];
                    }
                }
                syms.resize(w);
            }
        } else {
            for (int &s : syms)
                if (s < 0) s = t.unk_id;
        }
        if (cache.size() < (1u << 18)) cache.emplace(std::move(key), syms);
        out.insert(out.end(), syms.begin(), syms.end());
    }

    // pre-tokenize one stretch without special tokens
    void stretch(const unsigned char *s, size_t n, std::vector<int> &out)
    {
        size_t i = 0;
        while (i < n) {
            int l0;
            const unsigned c0 = decode(s, n, i, l0);
            // 1. contractions
            if (c0 == '\'' && i + 1 < n) {
                const unsigned char a = s[i + 1], b = i + 2 < n ? s[i + 2] : 0;
                int cl = 0;
                if (a == 's' || a == 't' || a == 'm' || a == 'd')
                    cl = 2;
                else if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e') || (a == 'l' && b == 'l'))
                    cl = 3;
                if (cl) {
                    bpe(s + i, cl, out);
                    i += cl;
                    continue;
                }
            }
            const uint8_t k0 = t.cls(c0);
            // 2-4.  " ?" + a run of letters / of numbers / of neither-space-nor-letter-nor-number
            size_t j = i;
            uint8_t kind = k0;
            if (c0 == ' ' && i + 1 < n) {
                int l1;
                const unsigned c1 = decode(s, n, i + 1, l1);
                const uint8_t k1 = t.cls(c1);
                if (k1 != SPACE) {
                    j = i + 1;
                    kind = k1;
                }
            }
            if (kind != SPACE) {
                size_t e = j;
                while (e < n) {
                    int l;
                    const unsigned c = decode(s, n, e, l);
                    if (t.cls(c) != kind) break;
                    e += l;
                }
                bpe(s + i, e - i, out);
                i = e;
                continue;
            }
            // 5-6.  a whitespace run: all of it at the end of the text; otherwise all but its last character (which the
            // next word's " ?" or the next \s+ takes) -- unless the run is a single character
            size_t e = i, last = i;
            while (e < n) {
                int l;
                const unsigned c = decode(s, n, e, l);
                if (t.cls(c) != SPACE) break;
                last = e;
                e += l;
            }
            if (e < n && last > i) e = last;
            bpe(s + i, e - i, out);
            i = e;
        }
    }

    void encode(const unsigned char *s, size_t n, std::vector<int> &out)
    {
        out.clear();
        size_t seg = 0, i = 0;
        if (!t.specials.empty()) {
            while (i < n) {
                const Special *hit = nullptr;
                for (const Special &sp : t.specials)
                    if (sp.text.size() <= n - i && s[i] == (unsigned char)sp.text[0] && memcmp(s + i, sp.text.data(), sp.text.size()) == 0 &&
                        (!hit || sp.text.size() > hit->text.size()))
                        hit = &sp;
                if (!hit) {
                    ++i;
                    continue;
                }
                size_t end = i;
                if (hit->lstrip) {   // the added token swallows the whitespace before it
                    while (end > seg) {
                        size_t b = end - 1;
                        while (b > seg && (s[b] & 0xC0) == 0x80) --b;
                        int l;
                        const unsigned c = decode(s, n, b, l);
                        if (t.cls(c) != SPACE) break;
                        end = b;
                    }
                }
                if (end > seg) stretch(s + seg, end - seg, out);
                out.push_back(hit->id);
                i += hit->text.size();
                seg = i;
            }
        }
        if (n > seg) stretch(s + seg, n - seg, out);
    }
};

}  // namespace

extern "C" {

struct crt_tokenizer;

// tokens[i] (UTF-8, NUL-terminated) has id ids[i]; merges are (left[i], right[i]) in rank order; specials are added tokens
// that are matched in the raw text (ids looked up in the vocabulary), lstrip[i] != 0 = swallows preceding whitespace.
crt_tokenizer *crt_create(int n_vocab, const char *const *tokens, const int32_t *ids, int n_merges, const char *const *left,
                          const char *const *right, int n_special, const char *const *specials, const int32_t *lstrip, const char *unk_token)
{
    Tokenizer *t = new Tokenizer();
    build_byte_alphabet(*t);
    int max_id = -1;
    for (int i = 0; i < n_vocab; ++i) {
        t->vocab.emplace(tokens[i], ids[i]);
        max_id = std::max(max_id, ids[i]);
    }
    t->id_to_sym.resize((size_t)max_id + 1);
    for (int i = 0; i < n_vocab; ++i) t->id_to_sym[ids[i]] = tokens[i];
    for (int b = 0; b < 256; ++b) {
        auto it = t->vocab.find(t->byte_sym[b]);
        t->byte_id[b] = it == t->vocab.end() ? -1 : it->second;
    }
    if (unk_token) {
        auto it = t->vocab.find(unk_token);
        if (it != t->vocab.end()) t->unk_id = it->second;
    }
    for (int b = 0; b < 256; ++b)
        if (t->byte_id[b] < 0 && t->unk_id < 0) {   // a byte the vocabulary cannot express and no <unk> to stand in: refuse
            delete t;
            return nullptr;
        }
    for (int i = 0; i < n_merges; ++i) {
        auto l = t->vocab.find(left[i]), r = t->vocab.find(right[i]);
        if (l == t->vocab.end() || r == t->vocab.end()) continue;
        auto m = t->vocab.find(std::string(left[i]) + right[i]);
        if (m == t->vocab.end()) continue;
        t->merges.emplace(((uint64_t)(uint32_t)l->second << 32) | (uint32_t)r->second, std::make_pair(i, m->second));
    }
    for (int i = 0; i < n_special; ++i) {
        auto it = t->vocab.find(specials[i]);
        if (it == t->vocab.end() || specials[i][0] == 0) continue;
        t->specials.push_back(Special{specials[i], it->second, lstrip && lstrip[i] != 0});
    }
    return reinterpret_cast<crt_tokenizer *>(t);
}

void crt_destroy(crt_tokenizer *h) { delete reinterpret_cast<Tokenizer *>(h); }

int crt_token_to_id(const crt_tokenizer *h, const char *token)
{
    const Tokenizer *t = reinterpret_cast<const Tokenizer *>(h);
    auto it = t->vocab.find(token);
    return it == t->vocab.end() ? -1 : it->second;
}

// texts[i] = UTF-8 bytes of length lens[i].  Row i of out_ids ([n, max_body] int32) receives the first max_body ids of
// text i (the rest of the row is left untouched), out_len[i] its TOTAL id count (may exceed max_body).  Returns the number of
// ids produced in total.  threads <= 0: one per hardware thread, at most 32.
int64_t crt_encode_batch(const crt_tokenizer *h, int64_t n, const char *const *texts, const int64_t *lens, int max_body, int32_t *out_ids,
                         int32_t *out_len, int threads)
{
    const Tokenizer *t = reinterpret_cast<const Tokenizer *>(h);
    if (!t || n <= 0) return 0;
    int nt = threads > 0 ? threads : (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency()));
    nt = (int)std::min<int64_t>(nt, n);
    std::atomic<int64_t> next(0), total(0);
    auto run = [&]() {
        Worker w(*t);
        std::vector<int> ids;
        int64_t mine = 0;
        for (;;) {
            const int64_t b0 = next.fetch_add(16);
            if (b0 >= n) break;
            for (int64_t i = b0; i < std::min<int64_t>(n, b0 + 16); ++i) {
                w.encode(reinterpret_cast<const unsigned char *>(texts[i]), (size_t)lens[i], ids);
                const int keep = (int)std::min<size_t>(ids.size(), (size_t)std::max(0, max_body));
                if (keep) memcpy(out_ids + (size_t)i * max_body, ids.data(), (size_t)keep * sizeof(int32_t));
                out_len[i] = (int32_t)ids.size();
                mine += (int64_t)ids.size();
            }
        }
        total += mine;
    };
    if (nt <= 1) {
        run();
    } else {
        std::vector<std::thread> pool;
        for (int k = 0; k < nt; ++k) pool.emplace_back(run);
        for (auto &th : pool) th.join();
    }
    return total.load();
}

// Number of matches of the pattern  \w+|[^\w\s]  (runs of word characters count once, every other non-space character once) in an
// ASCII text -- the stand-in token counter of the chunker (coderag_amd/indexer.py: CodeChunker counts cl100k tokens with tiktoken
// where that is installed; this image has no table for it).  Python's regex engine spends ~190 ns per match on it, which made the
// COUNT the largest single cost of indexing 14 k chunks through the reference-shaped surfaces (1.0 of 2.6 s).  Returns -1 at the
// first non-ASCII byte: the caller then counts with the Unicode-aware regex, so the result never depends on which path ran.
int64_t crt_count_wordish_ascii(const char *text, int64_t n)
{
    int64_t count = 0;
    bool in_word = false;
    for (int64_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)text[i];
        if (c >= 128) return -1;
        const bool word = (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_';
        if (word) {
            if (!in_word) ++count;
            in_word = true;
        } else {
            in_word = false;
            const bool space = c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31);   // str.isspace() over ASCII: what \s means in a str pattern
            if (!space) ++count;
        }
    }
    return count;
}

}  // extern "C"
