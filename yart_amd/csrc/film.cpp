// Film output: float framebuffer -> tone map -> 24-bit BMP, on the host.
//
// The reference's writer is Bitmap::write (src/bitmap.cpp:14-72) behind Camera::saveImage
// (include/camera.h:53-56).  Its tone map (src/bitmap.cpp:8-12) is kept operation for operation --
// expf, powf and the +0.5f truncation of the same libm -- and so is the file layout: 'BM', the
// 12-byte file header, the 40-byte info header (no resolution, no palette), rows bottom-up in B,G,R
// order.  What is fixed: the reference tests the padding ARRAY instead of the padding SIZE
// (`if (padding > 0)`, :64 -- rejected by g++ 11, always true where it compiles), which is
// harmless only because writing 0 bytes is; rows are padded to a multiple of four bytes here
// when, and only when, the width requires it.  Parity unpinned: src/bitmap.cpp does not compile
// with this container's g++, so no output of the reference's own writer exists to compare with
// (tests/test_film.py checks against the oracle's restatement and against the BMP format itself).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/yart_amd.h"

namespace yart {

static inline uint8_t toneMapChannel(float val)        // src/bitmap.cpp:8-12
{
    const float mapped = std::pow(1 - std::exp(-val), 1.0f / 2.2f) * 255 + 0.5f;
    if (!(mapped >= 0.0f)) return 0;                    // negative or NaN radiance: undefined in the reference
    return static_cast<uint8_t>(std::min(255.0f, mapped));
}

void toneMapBgr(const float* rgb, size_t nPixels, uint8_t* bgr)
{
    for (size_t i = 0; i < nPixels; ++i) {
        bgr[i * 3 + 0] = toneMapChannel(rgb[i * 3 + 2]);
        bgr[i * 3 + 1] = toneMapChannel(rgb[i * 3 + 1]);
        bgr[i * 3 + 2] = toneMapChannel(rgb[i * 3 + 0]);
    }
}

bool writeBmp(const char* path, const float* rgb, int32_t width, int32_t height, std::string& err)
{
    const int32_t rowSize = ((24 * width + 31) / 32) * 4;       // src/bitmap.cpp:18
    const uint32_t dataSize = static_cast<uint32_t>(rowSize) * static_cast<uint32_t>(height);
    uint8_t header[54];
    std::memset(header, 0, sizeof header);
    auto put32 = [&](int at, uint32_t v) { std::memcpy(header + at, &v, 4); };
    auto put16 = [&](int at, uint16_t v) { std::memcpy(header + at, &v, 2); };
    header[0] = 'B';
    header[1] = 'M';
    put32(2, 54u + dataSize);       // BitmapFileHeader::size
    put32(10, 54u);                 // ::offset
    put32(14, 40u);                 // BitmapInfoHeader::size
    put32(18, static_cast<uint32_t>(width));
    put32(22, static_cast<uint32_t>(height));
    put16(26, 1);                   // planes
    put16(28, 24);                  // bits per pixel
    put32(34, dataSize);
    FILE* f = std::fopen(path, "wb");
    if (!f) { err = std::string("cannot open [") + path + "] for writing"; return false; }
    bool ok = std::fwrite(header, 1, sizeof header, f) == sizeof header;
    std::vector<uint8_t> row(static_cast<size_t>(rowSize), 0);
    for (int32_t y = height - 1; y >= 0 && ok; --y) {           // bottom-up, src/bitmap.cpp:55
        toneMapBgr(rgb + static_cast<size_t>(y) * width * 3, static_cast<size_t>(width), row.data());
        ok = std::fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) err = std::string("short write to [") + path + "]";
    return ok;
}

} // namespace yart
