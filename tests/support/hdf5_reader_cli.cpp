// Test helper: prints what cmacionize_amd/host/Hdf5Reader.hpp reads from a
// file, as JSON.  usage: hdf5_reader_cli FILE PATH
#include "Hdf5Reader.hpp"

#include <cstdio>
#include <iostream>

int main(int argc, char **argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s FILE PATH\n", argv[0]);
    return 2;
  }
  try {
    cmi::Hdf5Reader file(argv[1]);
    const cmi::Hdf5Reader::Object o = file.open(argv[2]);
    std::printf("{\"members\": [");
    bool first = true;
    for (const auto &m : o.members) {
      std::printf("%s\"%s\"", first ? "" : ", ", m.first.c_str());
      first = false;
    }
    std::printf("], \"attributes\": {");
    first = true;
    for (const auto &a : o.attributes) {
      std::printf("%s\"%s\": ", first ? "" : ", ", a.first.c_str());
      first = false;
      if (a.second.type.cls == 3) {
        std::printf("\"%s\"", cmi::Hdf5Reader::as_string(a.second).c_str());
      } else {
        std::printf("[");
        const std::vector<double> v = cmi::Hdf5Reader::as_doubles(a.second);
        for (size_t i = 0; i < v.size(); ++i)
          std::printf("%s%.17g", i ? ", " : "", v[i]);
        std::printf("]");
      }
    }
    std::printf("}, \"dims\": [");
    for (size_t i = 0; i < o.dims.size(); ++i)
      std::printf("%s%llu", i ? ", " : "", (unsigned long long)o.dims[i]);
    std::printf("], \"layout\": %d, \"data\": [", o.layout);
    if (o.layout >= 0 && o.type.cls == 6) {
      /* a dataset of {name, value} records: "data": [], "dictionary": {} */
      std::printf("], \"dictionary\": {");
      bool first_entry = true;
      for (const auto &kv : file.read_dictionary(argv[2])) {
        std::printf("%s\"%s\": %.17g", first_entry ? "" : ", ",
                    kv.first.c_str(), kv.second);
        first_entry = false;
      }
      std::printf("}}\n");
      return 0;
    }
    if (o.layout >= 0) {
      const std::vector<double> v = file.read_doubles(argv[2]);
      for (size_t i = 0; i < v.size(); ++i)
        std::printf("%s%.17g", i ? ", " : "", v[i]);
    }
    std::printf("]}\n");
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
