#include "quantity.hpp"

#include <cstdlib>

namespace gangfit::host {

namespace {
constexpr i128 kLimit = (i128)1 << 126;

bool is_digit(char c) { return c >= '0' && c <= '9'; }

// suffix.go:108-198 — decimal SI, binary SI, decimal exponent
bool interpret_suffix(const std::string& s, int* base, int* exponent) {
    struct E {
        const char* s;
        int b, e;
    };
    static const E table[] = {{"", 10, 0},   {"n", 10, -9}, {"u", 10, -6}, {"m", 10, -3}, {"k", 10, 3},  {"M", 10, 6},
                              {"G", 10, 9},  {"T", 10, 12}, {"P", 10, 15}, {"E", 10, 18}, {"Ki", 2, 10}, {"Mi", 2, 20},
                              {"Gi", 2, 30}, {"Ti", 2, 40}, {"Pi", 2, 50}, {"Ei", 2, 60}};
    for (const E& e : table)
        if (s == e.s) {
            *base = e.b;
            *exponent = e.e;
            return true;
        }
    if (s.size() > 1 && (s[0] == 'e' || s[0] == 'E')) {  // strconv.ParseInt(suffix[1:], 10, 64)
        size_t i = 1;
        bool neg = false;
        if (s[i] == '+' || s[i] == '-') neg = s[i++] == '-';
        if (i >= s.size()) return false;
        long v = 0;
        for (; i < s.size(); ++i) {
            if (!is_digit(s[i])) return false;
            v = v * 10 + (s[i] - '0');
            if (v > 100000) return false;  // far outside anything representable; Go would build a huge inf.Dec
        }
        *base = 10;
        *exponent = (int)(neg ? -v : v);
        return true;
    }
    return false;
}
}  // namespace

bool Quantity::Parse(const std::string& str, Quantity* out) {
    if (str.empty()) return false;  // ErrFormatWrong
    size_t pos = 0, end = str.size();
    bool positive = true;
    if (str[0] == '-') {
        positive = false;
        ++pos;
    } else if (str[0] == '+') {
        ++pos;
    }
    while (pos < end && str[pos] == '0') ++pos;  // strip leading zeros
    size_t num_begin = pos;
    while (pos < end && is_digit(str[pos])) ++pos;
    std::string num = str.substr(num_begin, pos - num_begin), denom;
    if (pos < end && str[pos] == '.') {
        ++pos;
        size_t d0 = pos;
        while (pos < end && is_digit(str[pos])) ++pos;
        denom = str.substr(d0, pos - d0);
    }
    // suffix: letters of "eEinumkKMGTP", then an optional sign, then digits; anything left over is ErrFormatWrong
    size_t suffix_start = pos;
    while (pos < end && std::string("eEinumkKMGTP").find(str[pos]) != std::string::npos) ++pos;
    if (pos < end && (str[pos] == '-' || str[pos] == '+')) ++pos;
    while (pos < end && is_digit(str[pos])) ++pos;
    if (pos != end) return false;
    // the Go scanner needs at least one digit somewhere before the suffix ("" and "Gi" alone fail in ParseInt / SetString)
    if (num.empty() && denom.empty()) {
        bool had_zero = num_begin > (size_t)((str[0] == '-' || str[0] == '+') ? 1 : 0);
        if (!had_zero) return false;  // ErrNumeric
    }
    int base = 10, exponent = 0;
    if (!interpret_suffix(str.substr(suffix_start), &base, &exponent)) return false;  // ErrSuffix

    // |value| = digits * 10^(-len(denom)) * base^exponent ; nano = ceil(|value| * 10^9)
    i128 digits = 0;
    const std::string all = num + denom;
    for (char c : all) {
        if (digits > kLimit / 10) return false;
        digits = digits * 10 + (c - '0');
    }
    int scale10 = 9 - (int)denom.size();
    if (base == 10)
        scale10 += exponent;
    else
        for (int i = 0; i < exponent; ++i) {
            if (digits > kLimit / 2) return false;
            digits *= 2;
        }
    bool inexact = false;
    if (digits != 0) {
        for (; scale10 > 0; --scale10) {
            if (digits > kLimit / 10) return false;
            digits *= 10;
        }
        for (; scale10 < 0 && digits != 0; ++scale10) {
            if (digits % 10 != 0) inexact = true;
            digits /= 10;
        }
        if (inexact) digits += 1;  // inf.RoundUp: away from zero, to the nano (quantity.go:348-350)
    }
    // BinarySI values are capped at maxAllowed = 2^63 - 1 (quantity.go:353-356)
    if (base == 2) {
        const i128 cap = (i128)INT64_MAX * 1000000000;
        if (digits > cap) digits = cap;
    }
    *out = FromNano(positive ? digits : -digits);
    return true;
}

int64_t Quantity::Scaled(int64_t unit) const {  // ceil(|q| / unit) with the sign restored (quantity.go:744-755, math.go:169-199)
    const i128 a = nano_ < 0 ? -nano_ : nano_;
    i128 q = a / unit;
    if (a % unit != 0) q += 1;
    if (q > (i128)INT64_MAX) q = INT64_MAX;  // Go overflows silently; nothing on this path gets here (see canonical_*)
    return nano_ < 0 ? -(int64_t)q : (int64_t)q;
}

bool Quantity::Exact(int64_t unit, int64_t* out) const {
    if (nano_ % unit != 0) return false;
    const i128 q = nano_ / unit;
    const i128 lim = (i128)1 << 62;
    if (q >= lim || q <= -lim) return false;
    *out = (int64_t)q;
    return true;
}

}  // namespace gangfit::host
