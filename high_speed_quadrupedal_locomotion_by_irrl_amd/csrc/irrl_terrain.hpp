// irrl_terrain.hpp -- host-side generator of the height field used when `Terrain: True`.
//
// The reference asks RaiSim for a Perlin height map (Environment.hpp:254-264: frequency 1, zScale 0.1,
// 500 m x 20 m, 5000 x 500 samples, 3 octaves, lacunarity 2, gain 0.25, centred at the origin).  RaiSim's generator
// (seed, noise basis, coordinate scaling) is closed source, so the terrain itself is build-defined ("parity unpinned"
// like the rest of the physics): Ken Perlin's improved noise, permutation table shuffled by an LCG seeded with `seedd`,
// fractal sum  h(x,y) = zScale * sum_o gain^o * noise(frequency * lacunarity^o * (x, y))  with x, y in metres.
// The table is generated once on the host (double arithmetic, stored as f32) and shared by all robots of the pool;
// kernels sample it bilinearly and take the contact normal from the cell's gradient.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace irrl_host {

struct TerrainSpec {
  int nx = 5000, ny = 500;
  double x_size = 500.0, y_size = 20.0, z_scale = 0.1, frequency = 1.0, lacunarity = 2.0, gain = 0.25;
  int octaves = 3;
};

class PerlinNoise {
 public:
  explicit PerlinNoise(uint32_t seed) {
    int base[256];
    for (int i = 0; i < 256; i++) base[i] = i;
    uint32_t s = seed * 747796405u + 2891336453u;
    for (int i = 255; i > 0; i--) {  // Fisher-Yates with a 32-bit LCG
      s = s * 1664525u + 1013904223u;
      int j = (int)((s >> 8) % (uint32_t)(i + 1));
      int t = base[i]; base[i] = base[j]; base[j] = t;
    }
    for (int i = 0; i < 512; i++) p_[i] = base[i & 255];
  }
  double noise(double x, double y) const {  // improved noise, z = 0 slice
    const double z = 0.0;
    int X = (int)std::floor(x) & 255, Y = (int)std::floor(y) & 255, Z = 0;
    x -= std::floor(x); y -= std::floor(y);
    double u = fade(x), v = fade(y), w = fade(z);
    int A = p_[X] + Y, AA = p_[A] + Z, AB = p_[A + 1] + Z, B = p_[X + 1] + Y, BA = p_[B] + Z, BB = p_[B + 1] + Z;
    return lerp(w, lerp(v, lerp(u, grad(p_[AA], x, y, z), grad(p_[BA], x - 1, y, z)),
                        lerp(u, grad(p_[AB], x, y - 1, z), grad(p_[BB], x - 1, y - 1, z))),
                lerp(v, lerp(u, grad(p_[AA + 1], x, y, z - 1), grad(p_[BA + 1], x - 1, y, z - 1)),
                     lerp(u, grad(p_[AB + 1], x, y - 1, z - 1), grad(p_[BB + 1], x - 1, y - 1, z - 1))));
  }

 private:
  int p_[512];
  static double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
  static double lerp(double t, double a, double b) { return a + t * (b - a); }
  static double grad(int hash, double x, double y, double z) {
    int h = hash & 15;
    double u = h < 8 ? x : y, v = h < 4 ? y : ((h == 12 || h == 14) ? x : z);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
  }
};

inline void generate_heightfield(const TerrainSpec &t, uint32_t seed, std::vector<float> &out) {
  PerlinNoise pn(seed);
  out.resize((size_t)t.nx * t.ny);
  const double dx = t.x_size / (t.nx - 1), dy = t.y_size / (t.ny - 1);
  for (int i = 0; i < t.nx; i++) {
    const double x = -0.5 * t.x_size + i * dx;
    for (int j = 0; j < t.ny; j++) {
      const double y = -0.5 * t.y_size + j * dy;
      double f = t.frequency, a = 1.0, h = 0.0;
      for (int o = 0; o < t.octaves; o++) { h += a * pn.noise(x * f, y * f); f *= t.lacunarity; a *= t.gain; }
      out[(size_t)i * t.ny + j] = (float)(t.z_scale * h);
    }
  }
}

}  // namespace irrl_host
