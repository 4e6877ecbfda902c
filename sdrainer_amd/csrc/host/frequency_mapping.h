// frequency_mapping.h — host mirror of dsp.FrequencyMapping[F=int] (dsp/fft.go:95-135) and
// dsp.PeakCenterCorrection (dsp/fft.go:292-309).  Pure integer / float64 bookkeeping: stays on the
// host, exactly as the reference keeps it out of the per-bin loops.
#pragma once
#include <cmath>
#include <cstdint>

namespace host {

using BinLocation = double;          // dsp/fft.go:87
constexpr BinLocation BinFrom = -0.5;  // dsp/fft.go:89-93
constexpr BinLocation BinCenter = 0;
constexpr BinLocation BinTo = 0.5;

// Go's int(float64) on amd64 is CVTTSD2SQ: truncation toward zero; NaN and out-of-range values give
// the "integer indefinite" 0x8000000000000000.
inline int64_t go_int(double v)
{
    if (!(v > -9223372036854775808.0 && v < 9223372036854775808.0))
        return INT64_MIN;
    return (int64_t)v;
}

class FrequencyMapping {
public:
    FrequencyMapping(int sampleRate, int blockSize, int64_t centerFrequency)
        : sampleRate_(sampleRate), blockSize_(blockSize), binSize_((double)sampleRate / (double)blockSize),
          centerBin_(blockSize / 2)
    {
        SetCenterFrequency(centerFrequency);
    }
    void SetCenterFrequency(int64_t f)  // :119-122
    {
        centerFrequency_ = f;
        fromFrequency_ = f - sampleRate_ / 2;
    }
    int64_t BinToFrequency(int bin, BinLocation location) const  // :124-128
    {
        const double locationDelta = binSize_ * location;
        return (int64_t)((uint64_t)fromFrequency_ + (uint64_t)go_int((double)bin * binSize_ + locationDelta));
    }
    int FrequencyToBin(int64_t frequency) const  // :130-133
    {
        int64_t bin = go_int(((double)frequency - (double)fromFrequency_) / binSize_);
        if (bin > blockSize_ - 1)
            bin = blockSize_ - 1;
        if (bin < 0)
            bin = 0;
        return (int)bin;
    }
    int64_t CenterFrequency() const { return centerFrequency_; }
    int64_t FromFrequency() const { return fromFrequency_; }
    double BinSize() const { return binSize_; }

private:
    int sampleRate_, blockSize_;
    double binSize_;
    int centerBin_;
    int64_t centerFrequency_ = 0, fromFrequency_ = 0;
};

// dsp/fft.go:292-309 with the three neighbouring cumulation values already gathered
inline BinLocation PeakCenterCorrection(int bin, int blockSize, float c1, float c2, float c3)
{
    if (bin <= 0 || bin >= blockSize - 1)
        return 0;
    const double y1 = std::fabs((double)c1), y2 = std::fabs((double)c2), y3 = std::fabs((double)c3);
    return (y3 - y1) / (2 * (2 * y2 - y1 - y3));
}

}  // namespace host
