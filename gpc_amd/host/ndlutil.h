// ndlutil.h -- the reference's random-number utilities that the model code depends on (ndlutil.h:132-133,
// ndlutil.cpp:168-215, 267-375): a process-wide MT19937 seeded by init_genrand(seed) (CClctrl::setSeed), rand() =
// genrand_real3() in (0,1), and randpermTrunc().  std::mt19937 IS that generator (same seeding recurrence, same
// tempering), so seeded runs pick the same inducing points as the reference.
#ifndef GPC_AMD_NDLUTIL_H
#define GPC_AMD_NDLUTIL_H
#include <random>
#include <vector>

namespace ndlutil {
inline std::mt19937& generator()
{
  static std::mt19937 gen(5489u);   // genrand_int32's default seed when init_genrand was never called
  return gen;
}
inline void init_genrand(unsigned long s) { generator().seed((std::mt19937::result_type)(s & 0xffffffffUL)); }
inline double rand() { return ((double)generator()() + 0.5) * (1.0 / 4294967296.0); }   // genrand_real3
// the first `length` entries of a random permutation of 0..maxVal-1 (ndlutil.cpp:199-215)
inline std::vector<unsigned long> randpermTrunc(unsigned long maxVal, unsigned long length)
{
  std::vector<unsigned long> perm, indices;
  for(unsigned long i = 0; i < maxVal; i++) indices.push_back(i);
  for(unsigned long i = 0; i < length; i++) {
    const unsigned long ind = (unsigned long)(ndlutil::rand() * indices.size());
    perm.push_back(indices[ind]);
    indices.erase(indices.begin() + ind);
  }
  return perm;
}
}  // namespace ndlutil
#endif
