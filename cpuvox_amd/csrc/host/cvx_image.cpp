// cvx_image.cpp -- see cvx_image.h.
#include "cvx_image.h"

#include <zlib.h>

#include <algorithm>

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


// ---------------------------------------------------------------------------------------------------------------------
// JPEG (ITU T.81 / JFIF): 8-bit Huffman-coded baseline, extended-sequential and progressive frames (SOF0 / SOF1 / SOF2),
// 1 (grey) or 3 (YCbCr) components, any sampling factors, restart intervals.  Unity's Texture2D.LoadImage decodes
// JPEG map_Kd textures in the reference (SimpleMesh.cs:186-205); the engine's decoder is not part of /root/reference,
// so pixel values are "a correct JPEG decode": integer IDCT (the 13-bit "slow-but-accurate" factorisation), the IJG
// decoder's triangle-filter chroma upsampling, JFIF YCbCr -> RGB (within +-3 of libjpeg-turbo on the test images).  Arithmetic coding, 12-bit samples, CMYK: refused with an error.
// ---------------------------------------------------------------------------------------------------------------------
struct JpegHuffman {
	uint8_t bits[17] = {};
	uint8_t values[256] = {};
	int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
	bool defined = false;
	void Build()
	{
		int code = 0, k = 0;
		for (int l = 1; l <= 16; l++) {
			valptr[l] = k;
			mincode[l] = code;
			code += bits[l];
			k += bits[l];
			maxcode[l] = bits[l] ? code - 1 : -1;
			code <<= 1;
		}
		maxcode[17] = 0x7FFFFFFF;
		defined = true;
	}
};

struct JpegComponent {
	int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
	int blocksW = 0, blocksH = 0; // allocated blocks (whole MCUs)
	int dcPred = 0;
	std::vector<int16_t> coeff;   // blocksW * blocksH * 64, natural (de-zigzagged) order
	std::vector<uint8_t> plane;   // blocksW * 8 x blocksH * 8 samples after the IDCT
};

struct JpegBits {
	const uint8_t *p, *end;
	uint32_t acc = 0;
	int count = 0;
	bool marker = false; // ran into a marker: the entropy-coded segment is over (missing bits read as zeros)
	int Bit()
	{
		if (count == 0) {
			int b = 0;
			if (!marker && p < end) {
				b = *p++;
				if (b == 0xFF) {
					if (p < end && *p == 0x00) {
						p++;
					} else {
						marker = true;
						p--;
						b = 0;
					}
				}
			} else {
				marker = true;
			}
			acc = (uint32_t)b;
			count = 8;
		}
		count--;
		return (int)((acc >> count) & 1u);
	}
	int Receive(int n)
	{
		int v = 0;
		for (int i = 0; i < n; i++) { v = (v << 1) | Bit(); }
		return v;
	}
	static int Extend(int v, int n) { return n == 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); }
	int Decode(const JpegHuffman &t)
	{
		int code = 0;
		for (int l = 1; l <= 16; l++) {
			code = (code << 1) | Bit();
			if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) {
				return t.values[t.valptr[l] + code - t.mincode[l]];
			}
		}
		return -1;
	}
	void Reset() { count = 0; acc = 0; marker = false; }
};

const uint8_t kZigZag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
	                          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// Inverse DCT of one dequantised block, integer arithmetic (Loeffler-Ligtenberg-Moschytz factorisation, 13-bit constants,
// the structure of the IJG "islow" routine): identical results on every machine.
void JpegIdct(const int32_t *in, uint8_t *out, int stride)
{
	const int CONST_BITS = 13, PASS1_BITS = 2;
	const int32_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299,
	              F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
	int32_t ws[64];
	auto descale = [](int64_t x, int n) { return (int32_t)((x + ((int64_t)1 << (n - 1))) >> n); };
	for (int c = 0; c < 8; c++) { // columns
		const int32_t *p = in + c;
		if (!(p[8] | p[16] | p[24] | p[32] | p[40] | p[48] | p[56])) {
			const int32_t dc = p[0] * (1 << PASS1_BITS);
			for (int r = 0; r < 8; r++) { ws[r * 8 + c] = dc; }
			continue;
		}
		int64_t z2 = p[16], z3 = p[48];
		int64_t z1 = (z2 + z3) * F_0_541;
		int64_t tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
		z2 = p[0]; z3 = p[32];
		int64_t tmp0 = (z2 + z3) * (1 << CONST_BITS), tmp1 = (z2 - z3) * (1 << CONST_BITS);
		const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
		tmp0 = p[56]; tmp1 = p[40]; tmp2 = p[24]; tmp3 = p[8];
		z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
		int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * F_1_175;
		tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
		z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
		z3 += z5; z4 += z5;
		tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
		ws[0 * 8 + c] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
		ws[7 * 8 + c] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
		ws[1 * 8 + c] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
		ws[6 * 8 + c] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
		ws[2 * 8 + c] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
		ws[5 * 8 + c] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
		ws[3 * 8 + c] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
		ws[4 * 8 + c] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
	}
	for (int r = 0; r < 8; r++) { // rows
		const int32_t *p = ws + r * 8;
		int64_t z2 = p[2], z3 = p[6];
		int64_t z1 = (z2 + z3) * F_0_541;
		int64_t tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
		int64_t tmp0 = ((int64_t)p[0] + p[4]) * (1 << CONST_BITS), tmp1 = ((int64_t)p[0] - p[4]) * (1 << CONST_BITS);
		const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
		tmp0 = p[7]; tmp1 = p[5]; tmp2 = p[3]; tmp3 = p[1];
		z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
		int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * F_1_175;
		tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
		z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
		z3 += z5; z4 += z5;
		tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
		const int shift = CONST_BITS + PASS1_BITS + 3;
		const int64_t v[8] = { tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3 };
		for (int c = 0; c < 8; c++) {
			const int32_t s = descale(v[c], shift) + 128;
			out[r * stride + c] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
		}
	}
}

bool DecodeJpeg(const std::vector<uint8_t> &file, Image &out, std::string *error)
{
	uint16_t quant[4][64] = {};
	bool quantDefined[4] = {};
	JpegHuffman dcTables[4], acTables[4];
	std::vector<JpegComponent> comps;
	int width = 0, height = 0, hmax = 1, vmax = 1, mcusX = 0, mcusY = 0, restartInterval = 0;
	bool progressive = false, haveFrame = false, sawScan = false;
	size_t pos = 2;
	const size_t n = file.size();
	auto be16 = [&](size_t at) { return (int)(((unsigned)file[at] << 8) | file[at + 1]); };

	while (pos + 4 <= n) {
		if (file[pos] != 0xFF) { pos++; continue; }
		const int marker = file[pos + 1];
		if (marker == 0xFF || marker == 0x00) { pos++; continue; }
		if (marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01) { pos += 2; continue; }
		if (marker == 0xD9) { break; }
		const int length = be16(pos + 2);
		if (length < 2 || pos + 2 + (size_t)length > n) { return Fail(error, "truncated JPEG segment"); }
		const uint8_t *seg = file.data() + pos + 4; // (may be one past the end for an empty segment: never dereferenced then)
		const int segLen = length - 2;
		if (marker == 0xEE && segLen >= 12 && std::memcmp(seg, "Adobe", 5) == 0 && seg[11] != 1) {
			// APP14: transform 0 = RGB (or CMYK), 2 = YCCK.  Only JFIF-style YCbCr / grey is decoded; wrong colours without an error are worse than a refusal.
			return Fail(error, "unsupported JPEG colour transform (Adobe APP14: not YCbCr)");
		}
		if (marker == 0xDB) { // DQT
			int i = 0;
			while (i < segLen) {
				const int pq = seg[i] >> 4, tq = seg[i] & 15;
				i++;
				if (tq > 3 || i + (pq ? 128 : 64) > segLen) { return Fail(error, "bad JPEG quantisation table"); }
				for (int k = 0; k < 64; k++) {
					quant[tq][kZigZag[k]] = pq ? (uint16_t)((seg[i + 2 * k] << 8) | seg[i + 2 * k + 1]) : seg[i + k];
				}
				i += pq ? 128 : 64;
				quantDefined[tq] = true;
			}
		} else if (marker == 0xC4) { // DHT
			int i = 0;
			while (i + 17 <= segLen) {
				const int tc = seg[i] >> 4, th = seg[i] & 15;
				if (tc > 1 || th > 3) { return Fail(error, "bad JPEG Huffman table id"); }
				JpegHuffman &t = tc ? acTables[th] : dcTables[th];
				int total = 0;
				for (int l = 1; l <= 16; l++) { t.bits[l] = seg[i + l]; total += seg[i + l]; }
				i += 17;
				if (total > 256 || i + total > segLen) { return Fail(error, "bad JPEG Huffman table"); }
				std::memcpy(t.values, seg + i, (size_t)total);
				i += total;
				t.Build();
			}
		} else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) { // SOF0 / SOF1 / SOF2
			if (haveFrame) { return Fail(error, "JPEG with more than one frame"); }
			if (segLen < 6 || seg[0] != 8) { return Fail(error, "only 8-bit JPEG is supported"); }
			progressive = marker == 0xC2;
			height = be16(pos + 5);
			width = be16(pos + 7);
			const int nc = seg[5];
			if (width <= 0 || height <= 0 || (nc != 1 && nc != 3) || segLen < 6 + 3 * nc) { return Fail(error, "unsupported JPEG frame (grey and YCbCr only)"); }
			if ((int64_t)width * height > (int64_t)1 << 28) { return Fail(error, "JPEG too large"); }
			comps.resize((size_t)nc);
			for (int c = 0; c < nc; c++) {
				comps[(size_t)c].id = seg[6 + 3 * c];
				comps[(size_t)c].h = seg[7 + 3 * c] >> 4;
				comps[(size_t)c].v = seg[7 + 3 * c] & 15;
				comps[(size_t)c].tq = seg[8 + 3 * c];
				if (comps[(size_t)c].h < 1 || comps[(size_t)c].h > 4 || comps[(size_t)c].v < 1 || comps[(size_t)c].v > 4 || comps[(size_t)c].tq > 3) {
					return Fail(error, "bad JPEG component");
				}
				hmax = std::max(hmax, comps[(size_t)c].h);
				vmax = std::max(vmax, comps[(size_t)c].v);
			}
			mcusX = (width + 8 * hmax - 1) / (8 * hmax);
			mcusY = (height + 8 * vmax - 1) / (8 * vmax);
			for (JpegComponent &c : comps) {
				c.blocksW = mcusX * c.h;
				c.blocksH = mcusY * c.v;
				c.coeff.assign((size_t)c.blocksW * c.blocksH * 64, 0);
			}
			haveFrame = true;
		} else if (marker == 0xC3 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
			return Fail(error, "lossless / hierarchical / arithmetic-coded JPEG is not supported");
		} else if (marker == 0xDD) { // DRI
			if (segLen >= 2) { restartInterval = be16(pos + 4); }
		} else if (marker == 0xDA) { // SOS + entropy-coded data
			if (!haveFrame || segLen < 1) { return Fail(error, "JPEG scan before the frame header"); }
			const int ns = seg[0];
			if (ns < 1 || ns > (int)comps.size() || segLen < 1 + 2 * ns + 3) { return Fail(error, "bad JPEG scan header"); }
			std::vector<JpegComponent *> scan;
			for (int i = 0; i < ns; i++) {
				JpegComponent *found = nullptr;
				for (JpegComponent &c : comps) {
					if (c.id == seg[1 + 2 * i]) { found = &c; }
				}
				if (!found) { return Fail(error, "JPEG scan names an unknown component"); }
				found->td = seg[2 + 2 * i] >> 4;
				found->ta = seg[2 + 2 * i] & 15;
				if (found->td > 3 || found->ta > 3) { return Fail(error, "bad JPEG table selector"); }
				scan.push_back(found);
			}
			const int ss = seg[1 + 2 * ns], se = seg[2 + 2 * ns], ah = seg[3 + 2 * ns] >> 4, al = seg[3 + 2 * ns] & 15;
			if (!progressive && (ss != 0 || se != 63 || ah != 0 || al != 0)) { /* tolerated: some encoders write other values in sequential scans */ }
			if (progressive && (ss > se || se > 63 || (ss == 0 && se != 0) || al > 13)) { return Fail(error, "bad progressive JPEG scan parameters"); }
			for (JpegComponent *c : scan) {
				if ((!progressive || ss == 0) && ah == 0 && !dcTables[c->td].defined) { return Fail(error, "JPEG scan uses an undefined DC table"); }
				if ((!progressive || ss > 0) && !acTables[c->ta].defined && !(progressive && ss == 0)) { return Fail(error, "JPEG scan uses an undefined AC table"); }
				c->dcPred = 0;
			}
			JpegBits bits{ file.data() + pos + 2 + (size_t)length, file.data() + n };
			int eobrun = 0;
			// one block of a scan; returns false on a corrupt code
			auto decodeBlock = [&](JpegComponent &c, int16_t *blk) -> bool {
				if (!progressive) {
					const int t = bits.Decode(dcTables[c.td]);
					if (t < 0 || t > 11) { return false; }
					c.dcPred += JpegBits::Extend(bits.Receive(t), t);
					if (c.dcPred < -32768 || c.dcPred > 32767) { return false; } // (a crafted stream must not run the predictor into signed overflow)
					blk[0] = (int16_t)c.dcPred;
					for (int k = 1; k < 64;) {
						const int rs = bits.Decode(acTables[c.ta]);
						if (rs < 0) { return false; }
						const int r = rs >> 4, s = rs & 15;
						if (s == 0) {
							if (r != 15) { break; }
							k += 16;
							continue;
						}
						k += r;
						if (k > 63) { return false; }
						blk[kZigZag[k]] = (int16_t)JpegBits::Extend(bits.Receive(s), s);
						k++;
					}
					return true;
				}
				if (ss == 0) { // DC scan
					if (ah == 0) {
						const int t = bits.Decode(dcTables[c.td]);
						if (t < 0 || t > 11) { return false; }
						c.dcPred += JpegBits::Extend(bits.Receive(t), t);
						if (c.dcPred < -32768 || c.dcPred > 32767) { return false; }
						blk[0] = (int16_t)(c.dcPred * (1 << al));
					} else if (bits.Bit()) {
						blk[0] = (int16_t)(blk[0] | (1 << al));
					}
					return true;
				}
				if (ah == 0) { // AC first pass
					if (eobrun > 0) { eobrun--; return true; }
					for (int k = ss; k <= se;) {
						const int rs = bits.Decode(acTables[c.ta]);
						if (rs < 0) { return false; }
						const int r = rs >> 4, s = rs & 15;
						if (s == 0) {
							if (r < 15) {
								eobrun = (1 << r) - 1;
								if (r) { eobrun += bits.Receive(r); }
								break;
							}
							k += 16;
							continue;
						}
						k += r;
						if (k > 63) { return false; }
						blk[kZigZag[k]] = (int16_t)(JpegBits::Extend(bits.Receive(s), s) * (1 << al));
						k++;
					}
					return true;
				}
				// AC refinement
				const int p1 = 1 << al, m1 = -(1 << al);
				int k = ss;
				if (eobrun == 0) {
					for (; k <= se;) {
						const int rs = bits.Decode(acTables[c.ta]);
						if (rs < 0) { return false; }
						int r = rs >> 4;
						const int s = rs & 15;
						int value = 0;
						if (s == 0) {
							if (r < 15) {
								eobrun = 1 << r;
								if (r) { eobrun += bits.Receive(r); }
								break;
							}
						} else {
							if (s != 1) { return false; }
							value = bits.Bit() ? p1 : m1;
						}
						while (k <= se) {
							int16_t &coef = blk[kZigZag[k]];
							if (coef != 0) {
								if (bits.Bit() && (coef & p1) == 0) { coef = (int16_t)(coef >= 0 ? coef + p1 : coef + m1); }
							} else {
								if (r == 0) {
									if (value) { coef = (int16_t)value; }
									k++;
									break;
								}
								r--;
							}
							k++;
						}
					}
				}
				if (eobrun > 0) {
					for (; k <= se; k++) {
						int16_t &coef = blk[kZigZag[k]];
						if (coef != 0 && bits.Bit() && (coef & p1) == 0) { coef = (int16_t)(coef >= 0 ? coef + p1 : coef + m1); }
					}
					eobrun--;
				}
				return true;
			};
			int sinceRestart = 0;
			auto restartIfDue = [&]() {
				if (restartInterval > 0 && ++sinceRestart == restartInterval) {
					sinceRestart = 0;
					// skip to the RSTn marker, byte-align, reset predictors
					const uint8_t *q = bits.p;
					while (q + 1 < bits.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
						if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) { break; }
						q++;
					}
					if (q + 1 < bits.end && q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7) { q += 2; }
					bits.p = q;
					bits.Reset();
					eobrun = 0;
					for (JpegComponent *c : scan) { c->dcPred = 0; }
				}
			};
			bool ok = true;
			if (ns == 1) { // non-interleaved: the component's own blocks, only those that cover the image
				JpegComponent &c = *scan[0];
				const int bw = (((width * c.h + hmax - 1) / hmax) + 7) / 8, bh = (((height * c.v + vmax - 1) / vmax) + 7) / 8;
				for (int by = 0; by < bh && ok; by++) {
					for (int bx = 0; bx < bw && ok; bx++) {
						ok = decodeBlock(c, c.coeff.data() + ((size_t)by * c.blocksW + bx) * 64);
						restartIfDue();
					}
				}
			} else {
				for (int my = 0; my < mcusY && ok; my++) {
					for (int mx = 0; mx < mcusX && ok; mx++) {
						for (JpegComponent *c : scan) {
							for (int v = 0; v < c->v && ok; v++) {
								for (int h = 0; h < c->h && ok; h++) {
									ok = decodeBlock(*c, c->coeff.data() + ((size_t)(my * c->v + v) * c->blocksW + (mx * c->h + h)) * 64);
								}
							}
						}
						restartIfDue();
					}
				}
			}
			if (!ok) { return Fail(error, "corrupt JPEG entropy-coded data"); }
			sawScan = true;
			// continue behind the entropy-coded segment: at the next marker that is not a restart / stuffed byte
			const uint8_t *q = bits.p;
			while (q + 1 < bits.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) { q++; }
			pos = (size_t)(q - file.data());
			continue;
		}
		pos += 2 + (size_t)length;
	}
	if (!haveFrame || !sawScan) { return Fail(error, "JPEG without image data"); }

	for (JpegComponent &c : comps) {
		if (!quantDefined[c.tq]) { return Fail(error, "JPEG component uses an undefined quantisation table"); }
		const int stride = c.blocksW * 8;
		c.plane.assign((size_t)stride * c.blocksH * 8, 0);
		int32_t deq[64];
		for (int by = 0; by < c.blocksH; by++) {
			for (int bx = 0; bx < c.blocksW; bx++) {
				const int16_t *blk = c.coeff.data() + ((size_t)by * c.blocksW + bx) * 64;
				for (int k = 0; k < 64; k++) { deq[k] = (int32_t)blk[k] * (int32_t)quant[c.tq][k]; }
				JpegIdct(deq, c.plane.data() + (size_t)by * 8 * stride + (size_t)bx * 8, stride);
			}
		}
	}
	// Chroma upsampling to full resolution: the triangle filter of the IJG decoder ("fancy upsampling", the default of libjpeg
	// and libjpeg-turbo) for 2:1 horizontal and 2:1 x 2:1, sample replication for any other ratio.
	std::vector<std::vector<uint8_t>> full(comps.size());
	for (size_t ci = 0; ci < comps.size(); ci++) {
		const JpegComponent &c = comps[ci];
		const int stride = c.blocksW * 8;
		const int dw = (width * c.h + hmax - 1) / hmax, dh = (height * c.v + vmax - 1) / vmax; // downsampled size of the component
		std::vector<uint8_t> &dst = full[ci];
		dst.assign((size_t)width * height, 0);
		const bool h2 = hmax == 2 * c.h, v2 = vmax == 2 * c.v, h1 = hmax == c.h, v1 = vmax == c.v;
		if (h1 && v1) {
			for (int y = 0; y < height; y++) { std::memcpy(&dst[(size_t)y * width], &c.plane[(size_t)y * stride], (size_t)width); }
		} else if (h2 && (v1 || v2)) {
			std::vector<int> colsum((size_t)dw);
			for (int y = 0; y < height; y++) {
				const int sy = v2 ? y / 2 : y;
				// vertical: 3/4 of the nearer row + 1/4 of the farther one (edge rows are their own neighbours)
				int other = v2 ? ((y & 1) ? sy + 1 : sy - 1) : sy;
				other = other < 0 ? 0 : (other >= dh ? dh - 1 : other);
				const uint8_t *near = &c.plane[(size_t)sy * stride], *far = &c.plane[(size_t)other * stride];
				for (int x = 0; x < dw; x++) { colsum[(size_t)x] = v2 ? 3 * near[x] + far[x] : near[x]; }
				uint8_t *o = &dst[(size_t)y * width];
				for (int x = 0; x < width; x++) {
					const int sx = x / 2;
					const int cur = colsum[(size_t)sx];
					int value;
					if (v2) {
						if ((x & 1) == 0) { value = sx == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + colsum[(size_t)sx - 1] + 8) >> 4; }
						else { value = sx == dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + colsum[(size_t)sx + 1] + 7) >> 4; }
					} else {
						if ((x & 1) == 0) { value = sx == 0 ? cur : (cur * 3 + colsum[(size_t)sx - 1] + 1) >> 2; }
						else { value = sx == dw - 1 ? cur : (cur * 3 + colsum[(size_t)sx + 1] + 2) >> 2; }
					}
					o[x] = (uint8_t)value;
				}
			}
		} else {
			for (int y = 0; y < height; y++) {
				for (int x = 0; x < width; x++) { dst[(size_t)y * width + x] = c.plane[(size_t)(y * c.v / vmax) * stride + (x * c.h / hmax)]; }
			}
		}
	}
	out.width = width;
	out.height = height;
	out.rgba.assign((size_t)width * height * 4, 255);
	for (int y = 0; y < height; y++) {
		uint8_t *row = out.rgba.data() + (size_t)(height - 1 - y) * width * 4; // row 0 = bottom
		for (int x = 0; x < width; x++) {
			int s[3] = { 0, 128, 128 };
			for (size_t ci = 0; ci < comps.size(); ci++) { s[ci] = full[ci][(size_t)y * width + x]; }
			int r, g, b;
			if (comps.size() == 1) {
				r = g = b = s[0];
			} else { // JFIF: full-range BT.601, 16-bit fixed point
				const int cb = s[1] - 128, cr = s[2] - 128;
				r = s[0] + ((91881 * cr + 32768) >> 16);
				g = s[0] - ((22554 * cb + 46802 * cr + 32768) >> 16);
				b = s[0] + ((116130 * cb + 32768) >> 16);
			}
			row[4 * x + 0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
			row[4 * x + 1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
			row[4 * x + 2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
			row[4 * x + 3] = 255;
		}
	}
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
		ok = DecodeJpeg(file, out, &detail);
	} else if (file.size() >= 2 && file[0] == 'P' && file[1] == '6') {
		ok = DecodePpm(file, out, &detail);
	} else {
		ok = DecodeTga(file, out, &detail); // TGA has no signature
	}
	if (!ok) { return Fail(error, path + ": " + detail); }
	return true;
}

} // namespace cvx
