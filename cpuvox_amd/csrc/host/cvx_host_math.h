// cvx_host_math.h -- float vector / matrix helpers for the host-side mirror of
// the reference's managed code (RenderManager / UnityManager / CameraData ctor).
// Conventions restate UnityEngine + Unity.Mathematics behaviour (SURVEY.md
// Appendix B); these are NOT part of the bit-exact device contract, they only
// produce the inputs (SegmentData, CameraData) handed over the C ABI.
#pragma once

#include <cmath>
#include <cstdint>

namespace cvx {

struct float2 {
	float x = 0.f, y = 0.f;
	float2() = default;
	float2(float x_, float y_) : x(x_), y(y_) {}
	float &operator[](int i) { return i == 0 ? x : y; }
	float operator[](int i) const { return i == 0 ? x : y; }
};
inline float2 operator+(float2 a, float2 b) { return { a.x + b.x, a.y + b.y }; }
inline float2 operator-(float2 a, float2 b) { return { a.x - b.x, a.y - b.y }; }
inline float2 operator*(float2 a, float s) { return { a.x * s, a.y * s }; }
inline float2 operator/(float2 a, float2 b) { return { a.x / b.x, a.y / b.y }; }
inline float2 lerp(float2 a, float2 b, float t) { return a + (b - a) * t; }

struct float3 {
	float x = 0.f, y = 0.f, z = 0.f;
	float3() = default;
	float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
	float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
	float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline float3 operator+(float3 a, float3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline float3 operator-(float3 a, float3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float3 operator*(float3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline float3 operator/(float3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline float3 min3(float3 a, float3 b) { return { std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z) }; }
inline float3 max3(float3 a, float3 b) { return { std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z) }; }
// math.normalize = rsqrt(dot(x,x)) * x with rsqrt(x) = 1/sqrt(x)
inline float3 normalize(float3 a) { float r = 1.0f / std::sqrt(dot(a, a)); return a * r; }
inline float length(float3 a) { return std::sqrt(dot(a, a)); }

struct float4 {
	float x = 0.f, y = 0.f, z = 0.f, w = 0.f;
	float4() = default;
	float4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};

struct int3 {
	int x = 0, y = 0, z = 0;
};

// Column-major 4x4 like Unity.Mathematics.float4x4 / UnityEngine.Matrix4x4:
// m[c*4 + r].
struct mat4 {
	float m[16];
	float &at(int r, int c) { return m[c * 4 + r]; }
	float at(int r, int c) const { return m[c * 4 + r]; }
	static mat4 identity()
	{
		mat4 r{};
		for (int i = 0; i < 16; i++) { r.m[i] = 0.f; }
		r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f;
		return r;
	}
	static mat4 scale(float x, float y, float z)
	{
		mat4 r = identity();
		r.at(0, 0) = x; r.at(1, 1) = y; r.at(2, 2) = z;
		return r;
	}
	static mat4 translate(float x, float y, float z)
	{
		mat4 r = identity();
		r.at(0, 3) = x; r.at(1, 3) = y; r.at(2, 3) = z;
		return r;
	}
};

// math.mul(float4x4 a, float4 b) = a.c0*b.x + a.c1*b.y + a.c2*b.z + a.c3*b.w
inline float4 mul(const mat4 &a, float4 b)
{
	float4 r;
	r.x = a.m[0] * b.x + a.m[4] * b.y + a.m[8] * b.z + a.m[12] * b.w;
	r.y = a.m[1] * b.x + a.m[5] * b.y + a.m[9] * b.z + a.m[13] * b.w;
	r.z = a.m[2] * b.x + a.m[6] * b.y + a.m[10] * b.z + a.m[14] * b.w;
	r.w = a.m[3] * b.x + a.m[7] * b.y + a.m[11] * b.z + a.m[15] * b.w;
	return r;
}

inline mat4 mul(const mat4 &a, const mat4 &b)
{
	mat4 r;
	for (int c = 0; c < 4; c++) {
		float4 col = mul(a, float4(b.m[c * 4 + 0], b.m[c * 4 + 1], b.m[c * 4 + 2], b.m[c * 4 + 3]));
		r.m[c * 4 + 0] = col.x; r.m[c * 4 + 1] = col.y; r.m[c * 4 + 2] = col.z; r.m[c * 4 + 3] = col.w;
	}
	return r;
}

// General 4x4 inverse (cofactor expansion, float).
inline mat4 inverse(const mat4 &a)
{
	const float *m = a.m;
	float inv[16];
	inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
	inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
	inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
	inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
	inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
	inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
	inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
	inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
	inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
	inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
	inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
	inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
	inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
	inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
	inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
	inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
	float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
	float idet = 1.0f / det;
	mat4 r;
	for (int i = 0; i < 16; i++) { r.m[i] = inv[i] * idet; }
	return r;
}

// Mathf.RoundToInt = (int)Math.Round(f): half to even.
inline int RoundToInt(float f) { return (int)std::nearbyint(f); }
inline float Sign(float f) { return f >= 0.f ? 1.f : -1.f; } // Mathf.Sign(0) == 1
inline int NextPowerOfTwo(int v)
{
	if (v <= 1) { return v <= 0 ? 0 : 1; } // Mathf.NextPowerOfTwo(0) == 0, (1) == 1
	int p = 1;
	while (p < v) { p <<= 1; }
	return p;
}

constexpr float kDeg2Rad = 0.017453292519943295f;
constexpr float kRad2Deg = 57.29577951308232f;

} // namespace cvx
