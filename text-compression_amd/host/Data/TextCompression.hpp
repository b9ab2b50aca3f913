// TextCompression.hpp -- C++ host-side mirror of the reference's L3 module surface
// (Data.BWT / Data.MTF / Data.RLE / Data.FMIndex, ByteString instantiation) over the C ABI
// of include/textcomp.h.  Header only; links against libtextcomp.so.
//
// The reference is Haskell and no GHC exists in this image, so this is the compiled-language
// host layer: same function names, same argument meaning, same value shapes and the same
// error behaviour (what throws in the reference throws here), so that host code and tests
// read like the reference's.  Value shapes:
//   Seq (Maybe Word8)        -> std::vector<std::optional<uint8_t>>      (BWT Word8)
//   Seq (Maybe ByteString)   -> std::vector<std::optional<std::string>>  (RLE ByteString, BWT ByteString)
//   MTF ByteString           -> struct MTF { indices; final list }
//   Seq (ByteString, Maybe Int) -> std::vector<std::pair<std::string, std::optional<int64_t>>>
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "textcomp.h"

namespace Data {

struct TextCompError : std::runtime_error {
    int code;
    TextCompError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// one process-wide context on device 0 (tc_ctx = device + stream + workspace)
class Context {
  public:
    static tc_ctx *get() {
        static Context c;
        return c.ctx_;
    }
    static void check(int rc) {
        if (rc != TC_OK) throw TextCompError(rc, tc_last_error(get()));
    }

  private:
    Context() {
        int rc = tc_ctx_create(0, &ctx_);
        if (rc != TC_OK) throw TextCompError(rc, "tc_ctx_create: no usable HIP device (there is no CPU fallback)");
    }
    ~Context() { tc_ctx_destroy(ctx_); }
    tc_ctx *ctx_ = nullptr;
};

using Word8Seq = std::vector<std::optional<uint8_t>>;
using BSSeq = std::vector<std::optional<std::string>>;

namespace detail {
inline std::vector<int16_t> toSym(const Word8Seq &s) {
    std::vector<int16_t> v(s.size());
    for (size_t i = 0; i < s.size(); i++) v[i] = s[i] ? (int16_t)*s[i] : (int16_t)-1;
    return v;
}
inline Word8Seq fromSym(const std::vector<int16_t> &v) {
    Word8Seq s(v.size());
    for (size_t i = 0; i < v.size(); i++)
        if (v[i] >= 0) s[i] = (uint8_t)v[i];
    return s;
}
}  // namespace detail

namespace BWT {
// bytestringToBWT :: ByteString -> BWT Word8                              (BWT.hs:68-70)
inline Word8Seq bytestringToBWT(const std::string &bs) {
    if (bs.empty()) return {};  // BWT.hs:58
    std::vector<uint8_t> L(bs.size() + 1);
    uint64_t primary = 0;
    Context::check(tc_bwt_encode(Context::get(), (const uint8_t *)bs.data(), bs.size(), L.data(), &primary));
    Word8Seq out(L.size());
    for (size_t j = 0; j < L.size(); j++)
        if (j != primary) out[j] = L[j];
    return out;
}
// bytestringFromWord8BWT :: BWT Word8 -> ByteString                        (BWT.hs:108-110)
inline std::string bytestringFromWord8BWT(const Word8Seq &bwt) {
    if (bwt.empty()) return {};
    std::vector<int16_t> sym = detail::toSym(bwt);
    std::string out(bwt.size(), '\0');
    uint64_t n = 0;
    Context::check(tc_bwt_decode_sym(Context::get(), sym.data(), sym.size(), (uint8_t *)out.data(), &n));
    out.resize(n);
    return out;
}
}  // namespace BWT

namespace MTF {
struct MTFB {  // MTF ByteString = (Seq Int, Seq (Maybe ByteString))    (MTF/Internal.hs:67)
    std::vector<int> indices;
    BSSeq finalList;
    bool operator==(const MTFB &o) const { return indices == o.indices && finalList == o.finalList; }
};
// bytestringBWTToMTFB :: BWT Word8 -> MTF ByteString                       (MTF.hs:117-122)
inline MTFB bytestringBWTToMTFB(const Word8Seq &bwt) {
    MTFB out;
    if (bwt.empty()) return out;
    std::vector<int16_t> sym = detail::toSym(bwt);
    std::vector<uint16_t> idx(bwt.size());
    int16_t fl[TC_MAX_SIGMA];
    uint32_t sigma = 0;
    Context::check(tc_mtf_encode_sym(Context::get(), sym.data(), sym.size(), idx.data(), fl, &sigma));
    out.indices.assign(idx.begin(), idx.end());
    for (uint32_t i = 0; i < sigma; i++)
        out.finalList.push_back(fl[i] < 0 ? std::nullopt : std::optional<std::string>(std::string(1, (char)fl[i])));
    return out;
}
// bytestringToBWTToMTFB                                                   (MTF.hs:82-84)
inline MTFB bytestringToBWTToMTFB(const std::string &bs) { return bytestringBWTToMTFB(BWT::bytestringToBWT(bs)); }
// bytestringBWTFromMTFB :: MTF ByteString -> BWT ByteString               (MTF.hs:240-245)
inline Word8Seq bytestringBWTFromMTFB(const MTFB &m) {
    if (m.indices.empty() || m.finalList.empty()) return {};
    std::vector<uint16_t> idx(m.indices.begin(), m.indices.end());
    std::vector<int16_t> fl;
    for (auto &e : m.finalList) fl.push_back(e ? (int16_t)(uint8_t)(*e)[0] : (int16_t)-1);
    std::vector<int16_t> sym(idx.size());
    Context::check(tc_mtf_decode(Context::get(), idx.data(), idx.size(), fl.data(), (uint32_t)fl.size(), sym.data()));
    return detail::fromSym(sym);
}
// bytestringFromBWTFromMTFB                                               (MTF.hs:184-186)
inline std::string bytestringFromBWTFromMTFB(const MTFB &m) { return BWT::bytestringFromWord8BWT(bytestringBWTFromMTFB(m)); }
}  // namespace MTF

namespace RLE {
// bytestringBWTToRLEB :: BWT Word8 -> RLE ByteString                       (RLE.hs:117-123)
inline BSSeq bytestringBWTToRLEB(const Word8Seq &bwt) {
    BSSeq out;
    if (bwt.empty()) return out;  // RLE.hs:119
    std::vector<int16_t> sym = detail::toSym(bwt);
    uint64_t nruns = 2 * bwt.size() + 2;
    std::vector<uint32_t> counts(nruns);
    std::vector<int16_t> syms(nruns);
    Context::check(tc_rle_encode_sym(Context::get(), sym.data(), sym.size(), counts.data(), syms.data(), &nruns));
    for (uint64_t k = 0; k < nruns; k++) {
        out.push_back(std::to_string(counts[k]));  // `show count` (RLE/Internal.hs:128)
        out.push_back(syms[k] < 0 ? std::nullopt : std::optional<std::string>(std::string(1, (char)syms[k])));
    }
    return out;
}
// bytestringToBWTToRLEB                                                   (RLE.hs:83-85)
inline BSSeq bytestringToBWTToRLEB(const std::string &bs) { return bytestringBWTToRLEB(BWT::bytestringToBWT(bs)); }
// bytestringBWTFromRLEB :: RLE ByteString -> BWT ByteString               (RLE.hs:237-241)
inline Word8Seq bytestringBWTFromRLEB(const BSSeq &rle) {
    if (rle.empty()) return {};
    std::vector<uint32_t> counts;
    std::vector<int16_t> syms;
    for (size_t k = 0; k + 1 < rle.size(); k += 2) {  // a trailing odd element is ignored (:187-189)
        const auto &y1 = rle[k], &y2 = rle[k + 1];
        if (y1 && !y2) {
            counts.push_back(1);
            syms.push_back(-1);
            continue;
        }
        if (!y1 || !y2) throw TextCompError(TC_ERR_MALFORMED, "Maybe.fromJust: Nothing (RLE/Internal.hs:172-173)");
        size_t used = 0;
        long long c = 0;
        try {
            c = std::stoll(*y1, &used);
        } catch (...) {
            used = 0;
        }
        if (used != y1->size() || y1->empty()) throw TextCompError(TC_ERR_MALFORMED, "Prelude.read: no parse (RLE/Internal.hs:172)");
        counts.push_back(c > 0 ? (uint32_t)c : 0u);  // replicateM_ of a non-positive count is a no-op
        syms.push_back((int16_t)(uint8_t)(*y2)[0]);
    }
    uint64_t N = 1;
    for (size_t k = 0; k < counts.size(); k++) N += syms[k] < 0 ? 1 : counts[k];
    std::vector<int16_t> out(N);
    if (counts.empty()) return {};
    Context::check(tc_rle_decode(Context::get(), counts.data(), syms.data(), counts.size(), out.data(), &N));
    out.resize(N);
    return detail::fromSym(out);
}
// bytestringFromBWTFromRLEB                                               (RLE.hs:184-186)
inline std::string bytestringFromBWTFromRLEB(const BSSeq &rle) { return BWT::bytestringFromWord8BWT(bytestringBWTFromRLEB(rle)); }
}  // namespace RLE

namespace FMIndex {
using CountResult = std::vector<std::pair<std::string, std::optional<int64_t>>>;
// bytestringFMIndexCountS :: [ByteString] -> ByteString -> Seq (ByteString, Maybe Int)  (FMIndex.hs:362-379)
inline CountResult bytestringFMIndexCountS(const std::vector<std::string> &pats, const std::string &input) {
    CountResult out;
    if (pats.empty() || input.empty()) return out;  // FMIndex.hs:365-366
    tc_fm *fm = nullptr;
    Context::check(tc_fm_build(Context::get(), (const uint8_t *)input.data(), input.size(), &fm));
    std::string flat;
    std::vector<uint64_t> offs(1, 0);
    for (auto &p : pats) {
        flat += p;
        offs.push_back(flat.size());
    }
    flat.push_back('\0');
    std::vector<int64_t> counts(pats.size());
    int rc = tc_fm_count(Context::get(), fm, (const uint8_t *)flat.data(), offs.data(), pats.size(), counts.data());
    tc_fm_free(fm);
    Context::check(rc);
    for (size_t i = 0; i < pats.size(); i++)
        out.emplace_back(pats[i], counts[i] ? std::optional<int64_t>(counts[i]) : std::nullopt);
    return out;
}
// bytestringFMIndexCountP (FMIndex.hs:411-432): same values, same order; the spark pool over
// the pattern list is one batched launch.
inline CountResult bytestringFMIndexCountP(const std::vector<std::string> &pats, const std::string &input) {
    return bytestringFMIndexCountS(pats, input);
}
}  // namespace FMIndex

}  // namespace Data
