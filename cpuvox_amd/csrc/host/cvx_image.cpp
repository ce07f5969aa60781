// cvx_image.cpp -- see cvx_image.h.
#include "cvx_image.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace cvx {

namespace {

bool Fail(std::string *error, const std::string &text)
{
	if (error) { *error = text; }
	return false;
}

uint32_t Be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int Paeth(int a, int b, int c)
{
	const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
	return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

void FlipRows(Image &img)
{
	const size_t stride = (size_t)img.width * 4;
	std::vector<uint8_t> row(stride);
	for (int y = 0; y < img.height / 2; y++) {
		uint8_t *a = img.rgba.data() + (size_t)y * stride, *b = img.rgba.data() + (size_t)(img.height - 1 - y) * stride;
		std::memcpy(row.data(), a, stride);
		std::memcpy(a, b, stride);
		std::memcpy(b, row.data(), stride);
	}
}

// PNG (ISO/IEC 15948): non-interlaced, bit depth 8 or 16, colour types 0 2 3 4 6; tRNS for palettes.
bool DecodePng(const std::vector<uint8_t> &file, Image &out, std::string *error)
{
	size_t pos = 8;
	int width = 0, height = 0, depth = 0, colourType = 0, interlace = 0;
	std::vector<uint8_t> idat, palette, paletteAlpha;
	bool haveHeader = false;
	while (pos + 12 <= file.size()) {
		const uint32_t length = Be32(&file[pos]);
		const char *type = reinterpret_cast<const char *>(&file[pos + 4]);
		if (pos + 12 + (size_t)length > file.size()) { return Fail(error, "truncated PNG chunk"); }
		const uint8_t *data = &file[pos + 8];
		if (!std::memcmp(type, "IHDR", 4) && length >= 13) {
			width = (int)Be32(data);
			height = (int)Be32(data + 4);
			depth = data[8];
			colourType = data[9];
			interlace = data[12];
			haveHeader = true;
		} else if (!std::memcmp(type, "PLTE", 4)) {
			palette.assign(data, data + length);
		} else if (!std::memcmp(type, "tRNS", 4)) {
			paletteAlpha.assign(data, data + length);
		} else if (!std::memcmp(type, "IDAT", 4)) {
			idat.insert(idat.end(), data, data + length);
		} else if (!std::memcmp(type, "IEND", 4)) {
			break;
		}
		pos += 12 + (size_t)length;
	}
	if (!haveHeader || width <= 0 || height <= 0 || width > 32768 || height > 32768) { return Fail(error, "bad PNG header"); }
	if (interlace != 0) { return Fail(error, "interlaced PNG is not supported"); }
	if (depth != 8 && depth != 16) { return Fail(error, "PNG bit depth " + std::to_string(depth) + " is not supported"); }
	int channels;
	switch (colourType) {
	case 0: channels = 1; break;
	case 2: channels = 3; break;
	case 3: channels = 1; break;
	case 4: channels = 2; break;
	case 6: channels = 4; break;
	default: return Fail(error, "bad PNG colour type");
	}
	if (colourType == 3 && (depth != 8 || palette.empty())) { return Fail(error, "bad PNG palette"); }
	const size_t bpp = (size_t)channels * (size_t)(depth / 8), stride = bpp * (size_t)width;
	std::vector<uint8_t> raw((stride + 1) * (size_t)height);
	uLongf rawLength = (uLongf)raw.size();
	if (uncompress(raw.data(), &rawLength, idat.data(), (uLong)idat.size()) != Z_OK || rawLength != raw.size()) {
		return Fail(error, "PNG data does not inflate to the image size");
	}
	std::vector<uint8_t> prior(stride, 0), cur(stride);
	out.width = width;
	out.height = height;
	out.rgba.assign((size_t)width * height * 4, 255);
	for (int y = 0; y < height; y++) {
		const uint8_t *line = &raw[(size_t)y * (stride + 1)];
		const int filter = line[0];
		for (size_t i = 0; i < stride; i++) {
			const int a = i >= bpp ? cur[i - bpp] : 0, b = prior[i], c = i >= bpp ? prior[i - bpp] : 0;
			int v = line[1 + i];
			switch (filter) {
			case 0: break;
			case 1: v += a; break;
			case 2: v += b; break;
			case 3: v += (a + b) / 2; break;
			case 4: v += Paeth(a, b, c); break;
			default: return Fail(error, "bad PNG filter");
			}
			cur[i] = (uint8_t)v;
		}
		uint8_t *dst = &out.rgba[(size_t)y * width * 4];
		const size_t sample = (size_t)(depth / 8); // 16-bit samples: keep the high byte
		for (int x = 0; x < width; x++) {
			const uint8_t *p = &cur[(size_t)x * bpp];
			uint8_t *q = dst + (size_t)x * 4;
			switch (colourType) {
			case 0: q[0] = q[1] = q[2] = p[0]; break;
			case 2: q[0] = p[0]; q[1] = p[sample]; q[2] = p[2 * sample]; break;
			case 3: {
				const size_t idx = p[0];
				if (idx * 3 + 2 >= palette.size()) { return Fail(error, "PNG palette index out of range"); }
				q[0] = palette[idx * 3]; q[1] = palette[idx * 3 + 1]; q[2] = palette[idx * 3 + 2];
				q[3] = idx < paletteAlpha.size() ? paletteAlpha[idx] : 255;
				break;
			}
			case 4: q[0] = q[1] = q[2] = p[0]; q[3] = p[sample]; break;
			default: q[0] = p[0]; q[1] = p[sample]; q[2] = p[2 * sample]; q[3] = p[3 * sample]; break;
			}
		}
		prior.swap(cur);
	}
	FlipRows(out); // PNG rows run top-down
	return true;
}

// Truevision TGA, uncompressed true colour (type 2), 24 or 32 bits.
bool DecodeTga(const std::vector<uint8_t> &file, Image &out, std::string *error)
{
	if (file.size() < 18) { return Fail(error, "truncated TGA"); }
	const int idLength = file[0], mapType = file[1], type = file[2];
	const int width = file[12] | (file[13] << 8), height = file[14] | (file[15] << 8), bits = file[16], descriptor = file[17];
	if (mapType != 0 || type != 2 || (bits != 24 && bits != 32)) { return Fail(error, "only uncompressed 24/32-bit true-colour TGA is supported"); }
	const size_t bpp = (size_t)bits / 8, offset = 18 + (size_t)idLength;
	if (width <= 0 || height <= 0 || file.size() < offset + bpp * (size_t)width * height) { return Fail(error, "truncated TGA"); }
	out.width = width;
	out.height = height;
	out.rgba.resize((size_t)width * height * 4);
	for (size_t i = 0; i < (size_t)width * height; i++) {
		const uint8_t *p = &file[offset + i * bpp];
		uint8_t *q = &out.rgba[i * 4];
		q[0] = p[2]; q[1] = p[1]; q[2] = p[0]; q[3] = bpp == 4 ? p[3] : 255;
	}
	if (descriptor & 0x20) { FlipRows(out); } // bit 5: rows stored top-down
	return true;
}

// Netpbm P6 (binary RGB, maxval 255).
bool DecodePpm(const std::vector<uint8_t> &file, Image &out, std::string *error)
{
	size_t pos = 2;
	int values[3], n = 0;
	while (n < 3 && pos < file.size()) {
		if (file[pos] == '#') {
			while (pos < file.size() && file[pos] != '\n') { pos++; }
		} else if (file[pos] >= '0' && file[pos] <= '9') {
			int v = 0;
			while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') { v = v * 10 + (file[pos++] - '0'); }
			values[n++] = v;
			continue;
		}
		pos++;
	}
	if (n < 3 || values[2] != 255) { return Fail(error, "only P6 PPM with maxval 255 is supported"); }
	pos++; // the single whitespace byte after maxval
	const int width = values[0], height = values[1];
	if (width <= 0 || height <= 0 || file.size() < pos + (size_t)width * height * 3) { return Fail(error, "truncated PPM"); }
	out.width = width;
	out.height = height;
	out.rgba.resize((size_t)width * height * 4);
	for (size_t i = 0; i < (size_t)width * height; i++) {
		out.rgba[i * 4] = file[pos + i * 3];
		out.rgba[i * 4 + 1] = file[pos + i * 3 + 1];
		out.rgba[i * 4 + 2] = file[pos + i * 3 + 2];
		out.rgba[i * 4 + 3] = 255;
	}
	FlipRows(out);
	return true;
}

} // namespace

bool LoadImageFile(const std::string &path, Image &out, std::string *error)
{
	FILE *f = std::fopen(path.c_str(), "rb");
	if (!f) { return Fail(error, "cannot open " + path); }
	std::vector<uint8_t> file;
	uint8_t buffer[65536];
	size_t got;
	while ((got = std::fread(buffer, 1, sizeof buffer, f)) > 0) { file.insert(file.end(), buffer, buffer + got); }
	std::fclose(f);
	static const uint8_t pngSignature[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
	std::string detail;
	bool ok;
	if (file.size() >= 8 && !std::memcmp(file.data(), pngSignature, 8)) {
		ok = DecodePng(file, out, &detail);
	} else if (file.size() >= 3 && file[0] == 0xFF && file[1] == 0xD8) {
		ok = Fail(&detail, "JPEG textures are not supported by this build (Unity's Texture2D.LoadImage decodes them in the reference)");
	} else if (file.size() >= 2 && file[0] == 'P' && file[1] == '6') {
		ok = DecodePpm(file, out, &detail);
	} else {
		ok = DecodeTga(file, out, &detail); // TGA has no signature
	}
	if (!ok) { return Fail(error, path + ": " + detail); }
	return true;
}

} // namespace cvx
