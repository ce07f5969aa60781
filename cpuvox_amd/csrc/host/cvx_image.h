// cvx_image.h -- diffuse-texture loading for OBJ materials (map_Kd, SimpleMesh.cs:186-205).  The reference hands the
// file to UnityEngine's Texture2D.LoadImage (PNG / JPG); Unity is not available here, so PNG is decoded with zlib, JPEG
// (baseline, extended-sequential and progressive Huffman, grey / YCbCr) by the decoder in cvx_image.cpp, and uncompressed TGA
// and binary PPM are accepted as well.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace cvx {

struct Image {
	int width = 0, height = 0;
	std::vector<uint8_t> rgba; // row 0 = BOTTOM row, like Texture2D.GetPixels32 (SimpleMesh.cs:126)
};

bool LoadImageFile(const std::string &path, Image &out, std::string *error);

} // namespace cvx
