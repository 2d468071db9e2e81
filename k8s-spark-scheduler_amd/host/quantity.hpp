// quantity.hpp — resource.Quantity as the gang-fit path needs it (host side, C++ mirror of the Go types).
//
// Mirrors k8s.io/apimachinery v0.24.7 pkg/api/resource (vendored in the reference under vendor/k8s.io/apimachinery):
//   ParseQuantity        quantity.go:273-372   (scanner :147-270, suffixes suffix.go:108-198)
//   Value / MilliValue   quantity.go:731-755   ceil to whole units / milli-units, away from zero
//   Cmp / Add / Sub / Neg quantity.go:556-620
// A Quantity is held exactly as a signed 128-bit count of nano-units (ParseQuantity itself rounds anything finer than
// nano away from zero, quantity.go:343-350).  The device path works on canonical int64 (cpu milli, memory bytes, gpu
// devices — include/gangfit.h); `canonical_*` say whether a Quantity is exactly representable there.  The cgo shim must
// fall back to the Go packer when it is not (SURVEY.md section 8b "unit canonicalisation").
#pragma once

#include <cstdint>
#include <string>

namespace gangfit::host {

using i128 = __int128;

class Quantity {
public:
    Quantity() = default;
    static Quantity FromNano(i128 nano) {
        Quantity q;
        q.nano_ = nano;
        return q;
    }
    static Quantity FromInt(int64_t v) { return FromNano((i128)v * 1000000000); }    // resource.NewQuantity
    static Quantity FromMilli(int64_t v) { return FromNano((i128)v * 1000000); }     // resource.NewMilliQuantity

    // ParseQuantity: returns false (and leaves *out untouched) on ErrFormatWrong / ErrNumeric / ErrSuffix, and for
    // magnitudes beyond 2^96 units (the Go type would switch to inf.Dec; the shim treats those as non-representable).
    static bool Parse(const std::string& s, Quantity* out);

    int64_t Value() const { return Scaled(1000000000); }       // ceil away from zero to whole units
    int64_t MilliValue() const { return Scaled(1000000); }
    int Cmp(const Quantity& o) const { return nano_ < o.nano_ ? -1 : (nano_ > o.nano_ ? 1 : 0); }
    bool IsZero() const { return nano_ == 0; }
    void Add(const Quantity& o) { nano_ += o.nano_; }
    void Sub(const Quantity& o) { nano_ -= o.nano_; }
    void Neg() { nano_ = -nano_; }
    i128 nano() const { return nano_; }

    // exact canonical forms for the device tables; false when the value is not an exact multiple or |v| >= 2^62
    bool canonical_milli(int64_t* out) const { return Exact(1000000, out); }
    bool canonical_units(int64_t* out) const { return Exact(1000000000, out); }

private:
    int64_t Scaled(int64_t unit) const;
    bool Exact(int64_t unit, int64_t* out) const;
    i128 nano_ = 0;
};

}  // namespace gangfit::host
