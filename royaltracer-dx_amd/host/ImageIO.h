// ImageIO.h — image writers for the headless display path ("PNG/EXR writers").  The reference
// presents gOutput through a swap chain (Renderer.cpp:554-735) and has no file output; these replace the window.
//   PNG : 8-bit RGBA, zlib "stored" blocks (no compressor dependency), CRC-32 / Adler-32 computed here
//   EXR : OpenEXR 2 single-part scanline file, NO_COMPRESSION, three FLOAT channels B, G, R (linear radiance average)
//   PPM : binary P6
#pragma once
#include <cstdint>
#include <string>

bool WritePNG(const std::string& path, const uint8_t* rgba8, uint32_t width, uint32_t height);
bool WritePPM(const std::string& path, const uint8_t* rgba8, uint32_t width, uint32_t height);
// rgba32f: accumulation buffer (xyz = radiance sum, w = sample count) — written as xyz / max(w, 1)
bool WriteEXR(const std::string& path, const float* rgba32f, uint32_t width, uint32_t height);
