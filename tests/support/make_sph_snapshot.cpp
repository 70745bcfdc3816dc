/* Test support: writes a Gadget / SWIFT-like SPH snapshot (the input of
 * DensityFunction type GadgetSnapshot) with the host's own HDF5 writer.
 *   make_sph_snapshot OUT.hdf5 IN.bin PERIODIC BOXSIZE UL_CGS UM_CGS UT FLAGS
 * IN.bin: uint64 n, then doubles: coordinates [n][3], masses, smoothing
 * lengths, densities, temperatures, neutral fractions (n each).
 * FLAGS: bit 0 write /Units, bit 1 write Temperature, bit 2 write
 * NeutralFractionH, bit 3 write /RuntimePars. */
#include "Hdf5Writer.hpp"

#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>

int main(int argc, char **argv) {
  if (argc != 9) {
    std::cerr << "usage: make_sph_snapshot OUT IN PERIODIC BOX UL UM UT FLAGS\n";
    return 2;
  }
  std::ifstream in(argv[2], std::ios::binary);
  uint64_t n = 0;
  in.read(reinterpret_cast<char *>(&n), 8);
  auto block = [&](size_t count) {
    std::vector<double> v(count);
    in.read(reinterpret_cast<char *>(v.data()), 8 * count);
    return v;
  };
  const std::vector<double> coordinates = block(3 * n), masses = block(n),
                            h = block(n), density = block(n),
                            temperature = block(n), neutral = block(n);
  if (!in) {
    std::cerr << "short input\n";
    return 1;
  }
  const int periodic = std::atoi(argv[3]);
  const double box = std::atof(argv[4]);
  const int flags = std::atoi(argv[8]);
  cmi::Hdf5Writer file;
  file.attribute("Header", "BoxSize", std::vector<double>{box, box, box});
  if (flags & 8)
    file.attribute("RuntimePars", "PeriodicBoundariesOn", (int32_t)periodic);
  if (flags & 1) {
    file.attribute("Units", "Unit length in cgs (U_L)", std::atof(argv[5]));
    file.attribute("Units", "Unit mass in cgs (U_M)", std::atof(argv[6]));
    file.attribute("Units", "Unit temperature in cgs (U_T)",
                   std::atof(argv[7]));
  }
  file.dataset("PartType0", "Coordinates", {n, 3}, [&](std::ostream &os) {
    os.write(reinterpret_cast<const char *>(coordinates.data()),
             8 * coordinates.size());
  });
  file.dataset("PartType0", "Masses", masses);
  file.dataset("PartType0", "SmoothingLength", h);
  file.dataset("PartType0", "Density", density);
  if (flags & 2)
    file.dataset("PartType0", "Temperature", temperature);
  if (flags & 4)
    file.dataset("PartType0", "NeutralFractionH", neutral);
  file.write(argv[1]);
  return 0;
}
