"""Host-side BERT tokenisation (mirror of the reference's lxrt/tokenization.py:48-389 behaviour).

`BertTokenizer.tokenize` = basic tokenisation (clean, CJK spacing, lower-case + accent stripping, punctuation split;
reference :188-295) followed by greedy longest-match WordPiece (:298-348).  Results are memoised per input string:
the reference re-tokenises every sentence on every step (lxrt/entry.py:36-71, ~100 us per sentence), which would cap
the train step at ~10 k QA-pairs/s.  Pinned against reference-generated vectors in tests/golden/g4_*.
"""
import collections
import logging
import os
import unicodedata

logger = logging.getLogger(__name__)
VOCAB_NAME = "vocab.txt"
NEVER_SPLIT = ("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")


def load_vocab(vocab_file):
    """One wordpiece per line -> OrderedDict token -> index (reference :48-60)."""
    vocab = collections.OrderedDict()
    with open(vocab_file, "r", encoding="utf-8") as reader:
        for index, line in enumerate(reader):
            vocab[line.strip()] = index
    return vocab


def whitespace_tokenize(text):
    text = text.strip()
    return text.split() if text else []


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    return ch not in "\t\n\r" and unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class BasicTokenizer(object):
    def __init__(self, do_lower_case=True, never_split=NEVER_SPLIT):
        self.do_lower_case = do_lower_case
        self.never_split = never_split

    def tokenize(self, text):
        buf = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                buf.append(" ")
            elif _is_cjk(cp):
                buf.extend((" ", ch, " "))
            else:
                buf.append(ch)
        out = []
        for tok in "".join(buf).split():
            if self.do_lower_case and tok not in self.never_split:
                tok = tok.lower()
                tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
            if tok in self.never_split:
                out.append(tok)
                continue
            word = []
            for ch in tok:
                if _is_punctuation(ch):
                    if word:
                        out.append("".join(word))
                        word = []
                    out.append(ch)
                else:
                    word.append(ch)
            if word:
                out.append("".join(word))
        return whitespace_tokenize(" ".join(out))


class WordpieceTokenizer(object):
    def __init__(self, vocab, unk_token="[UNK]", max_input_chars_per_word=100):
        self.vocab, self.unk_token, self.max_input_chars_per_word = vocab, unk_token, max_input_chars_per_word

    def tokenize(self, text):
        out = []
        for token in whitespace_tokenize(text):
            if len(token) > self.max_input_chars_per_word:
                out.append(self.unk_token)
                continue
            pieces, start, bad = [], 0, False
            while start < len(token):
                end, cur = len(token), None
                while start < end:
                    sub = token[start:end] if start == 0 else "##" + token[start:end]
                    if sub in self.vocab:
                        cur = sub
                        break
                    end -= 1
                if cur is None:
                    bad = True
                    break
                pieces.append(cur)
                start = end
            out.extend([self.unk_token] if bad else pieces)
        return out


class BertTokenizer(object):
    """Punctuation splitting + wordpiece, same constructor and methods as the reference class (:72-133)."""

    def __init__(self, vocab_file, do_lower_case=True, max_len=None, do_basic_tokenize=True, never_split=NEVER_SPLIT):
        if not os.path.isfile(vocab_file):
            raise ValueError("Can't find a vocabulary file at path '{}'.".format(vocab_file))
        self.vocab = load_vocab(vocab_file)
        self.vocab_file = vocab_file
        self.do_lower_case = do_lower_case
        self.ids_to_tokens = collections.OrderedDict((i, t) for t, i in self.vocab.items())
        self.do_basic_tokenize = do_basic_tokenize
        if do_basic_tokenize:
            self.basic_tokenizer = BasicTokenizer(do_lower_case=do_lower_case, never_split=never_split)
        self.wordpiece_tokenizer = WordpieceTokenizer(vocab=self.vocab)
        self.max_len = max_len if max_len is not None else int(1e12)
        self._cache = {}

    def tokenize(self, text):
        hit = self._cache.get(text)
        if hit is not None:
            return list(hit)
        if self.do_basic_tokenize:
            toks = [sub for tok in self.basic_tokenizer.tokenize(text) for sub in self.wordpiece_tokenizer.tokenize(tok)]
        else:
            toks = self.wordpiece_tokenizer.tokenize(text)
        if len(self._cache) < 1000000:
            self._cache[text] = tuple(toks)
        return toks

    def convert_tokens_to_ids(self, tokens):
        ids = [self.vocab[t] for t in tokens]
        if len(ids) > self.max_len:
            logger.warning("Token indices sequence length is longer than the specified maximum sequence length "
                           "for this BERT model ({} > {}).".format(len(ids), self.max_len))
        return ids

    def convert_ids_to_tokens(self, ids):
        return [self.ids_to_tokens[i] for i in ids]

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, cache_dir=None, *inputs, **kwargs):
        """Offline resolution of the vocabulary (the reference downloads it, :135-171): a file or directory path, else
        $RGQA_BERT_VOCAB, else $RGQA_BERT_DIR/vocab.txt, else snap/bert/vocab.txt. Returns None when nothing is found,
        like the reference does on a failed download."""
        cands = [pretrained_model_name_or_path, os.environ.get("RGQA_BERT_VOCAB", ""),
                 os.path.join(os.environ.get("RGQA_BERT_DIR", ""), VOCAB_NAME), os.path.join("snap", "bert", VOCAB_NAME)]
        for c in cands:
            if c and os.path.isdir(c):
                c = os.path.join(c, VOCAB_NAME)
            if c and os.path.isfile(c):
                if pretrained_model_name_or_path.startswith("bert-"):
                    kwargs["max_len"] = min(kwargs.get("max_len", int(1e12)), 512)
                return cls(c, *inputs, **kwargs)
        logger.error("Vocabulary for '%s' not found (no network here): set RGQA_BERT_VOCAB to a vocab.txt",
                     pretrained_model_name_or_path)
        return None


class NativeBatchEncoder(object):
    """Batch front end of the native tokenizer in librgqa_hip.so (`rgqa_tokenizer_*`, csrc/tokenizer.hip): one call turns a
    list of sentences into the `[n, T]` id / mask arrays and the token counts that `convert_sents_to_features` (reference
    lxrt/entry.py:36-71) builds sentence by sentence in Python (~100 us each there, ~1 us here). Sentences with non-ASCII
    characters or an embedded NUL are reported back (`needs_python`) for the Python implementation of the same rules."""

    def __init__(self, tokenizer):
        import ctypes as C
        from .. import _lib
        self._C, self._lib = C, _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.rgqa_tokenizer_create(tokenizer.vocab_file.encode("utf-8"), 1 if tokenizer.do_lower_case else 0, C.byref(self._h)))
        n = C.c_int64()
        _lib.check(self._lib.rgqa_tokenizer_vocab_size(self._h, C.byref(n)))
        if n.value != len(tokenizer.vocab):
            raise RuntimeError("native tokenizer read %d vocabulary entries, Python read %d" % (n.value, len(tokenizer.vocab)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.rgqa_tokenizer_destroy(h)

    def encode(self, sents, max_seq_length, ids, mask):
        """ids, mask: writable contiguous int64 numpy arrays [n, T] (e.g. views of pinned tensors). Returns (lengths int32[n],
        needs_python uint8[n])."""
        import numpy as np
        C = self._C
        n = len(sents)
        raw = [s.encode("utf-8") for s in sents]
        arr = (C.c_char_p * n)(*raw)
        lengths = np.zeros(n, dtype=np.int32)
        needs = np.zeros(n, dtype=np.uint8)
        from .. import _lib
        _lib.check(self._lib.rgqa_tokenizer_encode(self._h, arr, n, max_seq_length, C.c_void_p(ids.ctypes.data), C.c_void_p(mask.ctypes.data),
                                                   C.c_void_p(lengths.ctypes.data), C.c_void_p(needs.ctypes.data)))
        for i, b in enumerate(raw):
            if b"\x00" in b:
                needs[i] = 1           # a C string ends at NUL; the reference drops the NUL and keeps going
        return lengths, needs
