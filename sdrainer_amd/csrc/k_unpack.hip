// k_unpack.hip — source-side wire format on the device: KiwiSDR "SND" frames carry
// IQ as big-endian int16 pairs (kiwi/client.go:284-308 decodeIQMessage / decodeIQBytes:
// float32(int16(be16)) / float32(math.MaxInt16)).  Unpacking in HBM halves the PCIe bytes of that
// source: the raw payload is what gets uploaded.
#include <hip/hip_runtime.h>

#include "sdr_device.h"

namespace sdr {

__device__ __forceinline__ float be16_to_f32(uint32_t lo_byte_first)
{
    // bytes b0 b1 (network order): value = int16(b0 << 8 | b1)
    const uint32_t v = ((lo_byte_first & 0xffu) << 8) | ((lo_byte_first >> 8) & 0xffu);
    return __fdiv_rn((float)(int16_t)v, 32767.0f);  // correctly rounded float32 division, as Go's
}

// 8 payload bytes -> 4 floats per thread (16-byte stores)
__global__ __launch_bounds__(256) void k_unpack_be16(const uint8_t *__restrict__ raw, float *__restrict__ out,
                                                     size_t n_values)
{
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = i4 * 4;
    if (i >= n_values)
        return;
    if (i + 4 <= n_values) {
        const uint2 w = *reinterpret_cast<const uint2 *>(raw + 2 * i);  // staging buffers are 8-byte aligned
        float4 f;
        f.x = be16_to_f32(w.x & 0xffffu);
        f.y = be16_to_f32(w.x >> 16);
        f.z = be16_to_f32(w.y & 0xffffu);
        f.w = be16_to_f32(w.y >> 16);
        *reinterpret_cast<float4 *>(out + i) = f;
    } else {
        for (size_t k = i; k < n_values; k++)
            out[k] = be16_to_f32((uint32_t)raw[2 * k] | ((uint32_t)raw[2 * k + 1] << 8));
    }
}

hipError_t launch_unpack_be16(const uint8_t *raw, float *out, size_t n_values, hipStream_t stream)
{
    if (n_values == 0)
        return hipSuccess;
    const size_t threads = (n_values + 3) / 4;
    hipLaunchKernelGGL(k_unpack_be16, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, raw, out, n_values);
    return hipGetLastError();
}

}  // namespace sdr
