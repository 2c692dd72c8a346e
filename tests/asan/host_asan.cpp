// AddressSanitizer / UBSan build of the library's HOST translation units (tokenizer.hip, hostio.hip, runtime.hip: no device code),
// compiled as plain C++ with the ROCm clang and driven through the C ABI on the golden tokenizer fixture (tests/golden/g4_*).
// Built and run by tests/test_host.py::test_host_units_under_address_sanitizer (CPU only; GPU ASan is not available on this pool).
//   usage: host_asan <vocab.txt> <sentences.json-lines file: one sentence per line> <max_seq_length>
// prints one line per sentence: ids..., then "gather ok".
#include "../../rgqa_amd/csrc/runtime.hip"
#include "../../rgqa_amd/csrc/tokenizer.hip"
#include "../../rgqa_amd/csrc/hostio.hip"
#include <fstream>
#include <iostream>

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    rgqa_tokenizer* t = nullptr;
    if (rgqa_tokenizer_create(argv[1], 1, &t) != 0) { fprintf(stderr, "%s\n", rgqa_last_error_string()); return 1; }
    rgqa_tokenizer* bad = nullptr;
    if (rgqa_tokenizer_create("/nonexistent/vocab.txt", 1, &bad) == 0) return 1;       // error path: message, no leak
    std::vector<std::string> sents;
    std::ifstream in(argv[2]);
    for (std::string l; std::getline(in, l);) sents.push_back(l);
    sents.push_back("");                                          // empty sentence
    sents.push_back(std::string(300, 'a'));                       // one word longer than max_input_chars_per_word
    sents.push_back(std::string("tab\there \x01\x02 ctrl ?!?!") + std::string(400, '?'));   // control characters, far more tokens than T
    const int T = atoi(argv[3]), n = (int)sents.size();
    std::vector<const char*> ptrs;
    for (auto& s : sents) ptrs.push_back(s.c_str());
    std::vector<int64_t> ids((size_t)n * T), mask((size_t)n * T);
    std::vector<int32_t> len(n);
    std::vector<uint8_t> py(n);
    if (rgqa_tokenizer_encode(t, ptrs.data(), n, T, ids.data(), mask.data(), len.data(), py.data()) != 0) { fprintf(stderr, "%s\n", rgqa_last_error_string()); return 1; }
    for (int i = 0; i < n; ++i) {
        printf("%d %d", (int)py[i], (int)len[i]);
        for (int k = 0; k < T; ++k) printf(" %lld", (long long)ids[(size_t)i * T + k]);
        printf("\n");
    }
    if (rgqa_tokenizer_encode(t, ptrs.data(), n, 1, ids.data(), mask.data(), len.data(), py.data()) == 0) return 1;      // T < 2 must be refused
    int64_t vs = 0;
    rgqa_tokenizer_vocab_size(t, &vs);
    rgqa_tokenizer_destroy(t);
    // host row gather: 7 rows of 1000 bytes, 4 threads, every destination byte checked; out-of-range row refused
    std::vector<unsigned char> src(7 * 1000), dst(5 * 1000);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)(i * 31 + 7);
    const int64_t rows[5] = {6, 0, 3, 3, 1};
    if (rgqa_host_gather_rows(src.data(), 1000, 7, rows, 5, dst.data(), 4) != 0) return 1;
    for (int i = 0; i < 5; ++i)
        if (memcmp(dst.data() + i * 1000, src.data() + rows[i] * 1000, 1000) != 0) return 1;
    const int64_t badrow[1] = {7};
    if (rgqa_host_gather_rows(src.data(), 1000, 7, badrow, 1, dst.data(), 2) == 0) return 1;
    printf("gather ok %lld\n", (long long)vs);
    return 0;
}
