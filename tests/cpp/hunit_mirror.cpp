// The reference's HUnit cases (RLE.hs:313-320, MTF.hs:287-299) and the documented FM-index
// example, run through the C++ host mirror (Data/TextCompression.hpp) on the HIP library.
// Vectors come from a line-based dump of tests/golden/hunit_vectors.json written by the
// pytest wrapper (tests/test_gpu_cpp_mirror.py); nothing is hard-coded here.
#include <cstdio>
#include <fstream>
#include <iostream>

#include "Data/TextCompression.hpp"

using namespace Data;

static std::optional<std::string> elem(std::ifstream &f) {
    std::string line;
    std::getline(f, line);
    if (line == "N") return std::nullopt;
    return line.substr(2);
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1]);
    std::string kind;
    int failures = 0, cases = 0;
    auto expect = [&](bool ok, const char *name) {
        cases++;
        if (!ok) {
            failures++;
            std::printf("FAIL %s\n", name);
        }
    };
    while (std::getline(f, kind)) {
        if (kind == "rle") {
            std::string text, cnt;
            std::getline(f, text);
            std::getline(f, cnt);
            BSSeq want;
            for (int i = 0, k = std::stoi(cnt); i < k; i++) want.push_back(elem(f));
            expect(RLE::bytestringToBWTToRLEB(text) == want, "assertEqual rleK (textToBWTToRLEB sK)");
            expect(RLE::bytestringFromBWTFromRLEB(want) == text, "assertEqual sK (textFromBWTFromRLEB rleK)");
        } else if (kind == "mtf") {
            std::string text, cnt;
            std::getline(f, text);
            std::getline(f, cnt);
            MTF::MTFB want;
            for (int i = 0, k = std::stoi(cnt); i < k; i++) {
                std::string v;
                std::getline(f, v);
                want.indices.push_back(std::stoi(v));
            }
            std::getline(f, cnt);
            for (int i = 0, k = std::stoi(cnt); i < k; i++) want.finalList.push_back(elem(f));
            expect(MTF::bytestringToBWTToMTFB(text) == want, "assertEqual MTF (textToBWTToMTFB)");
            expect(MTF::bytestringFromBWTFromMTFB(want) == text, "assertEqual text (textFromBWTFromMTFB)");
        } else if (kind == "count") {
            std::string text, cnt;
            std::getline(f, text);
            std::getline(f, cnt);
            std::vector<std::string> pats;
            std::vector<long long> want;
            for (int i = 0, k = std::stoi(cnt); i < k; i++) {
                std::string p, c;
                std::getline(f, p);
                std::getline(f, c);
                pats.push_back(p);
                want.push_back(std::stoll(c));
            }
            auto got = FMIndex::bytestringFMIndexCountS(pats, text);
            bool ok = got.size() == pats.size();
            for (size_t i = 0; ok && i < got.size(); i++)
                ok = got[i].first == pats[i] && (want[i] ? got[i].second == std::optional<int64_t>(want[i]) : !got[i].second);
            expect(ok, "bytestringFMIndexCountS");
        }
    }
    // error behaviour mirrors the reference: what throws there throws here
    try {
        RLE::bytestringBWTFromRLEB({std::string("x"), std::string("a")});
        expect(false, "read: no parse must throw");
    } catch (const TextCompError &e) {
        expect(e.code == TC_ERR_MALFORMED, "read: no parse -> TC_ERR_MALFORMED");
    }
    expect(BWT::bytestringToBWT("").empty(), "toBWT [] = BWT Empty");
    std::printf("%d cases, %d failures\n", cases, failures);
    return failures ? 1 : 0;
}
