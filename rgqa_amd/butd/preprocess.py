"""Mirror of the reference's butd/preprocess.py `Dictionary` (14-51): lower-case, strip , . ?, split "'s", unknown words
map to the padding index (= ntoken)."""


class Dictionary(object):
    def __init__(self, word2idx=None, idx2word=None):
        self.word2idx = {} if word2idx is None else word2idx
        self.idx2word = [] if idx2word is None else idx2word

    @property
    def ntoken(self):
        return len(self.word2idx)

    @property
    def padding_idx(self):
        return len(self.word2idx)

    def tokenize(self, sentence, add_word):
        sentence = sentence.lower().replace(",", "").replace(".", "").replace("?", "").replace("'s", " 's")
        words = sentence.split()
        if add_word:
            return [self.add_word(w) for w in words]
        pad = self.padding_idx
        return [self.word2idx.get(w, pad) for w in words]

    def add_word(self, word):
        if word not in self.word2idx:
            self.idx2word.append(word)
            self.word2idx[word] = len(self.idx2word) - 1
        return self.word2idx[word]

    def __len__(self):
        return len(self.idx2word)
