// Host-side harness: plonky_amd/csrc/hostnorm.cpp (the host normalisation of the points that cross PCIe, plain C++) behind a C symbol,
// so that tests/test_hostnorm.py can sweep it against Python integers without a GPU.
#include "../plonky_amd/csrc/hostnorm.cpp"

extern "C" int hostnorm_to_affine(int curve, unsigned count, const uint8_t* xyz, const uint8_t* zero, uint8_t* xy) {
    return plk::host_projective_to_affine(curve, count, xyz, zero, xy);
}
