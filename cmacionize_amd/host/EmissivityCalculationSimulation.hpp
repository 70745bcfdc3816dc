/*
 * EmissivityCalculationSimulation.hpp - the reference's emission mode
 * (`CMacIonize --emission --params lines.param --file snapshot.hdf5`,
 * src/EmissivityCalculationSimulation.cpp:58-299) on the GPU engine: reads the
 * state of a snapshot, computes the emission lines flagged in
 * "EmissivityValues:<name>" on the device (cmi_gpu_compute_emissivities) and
 * adds them to the snapshot as datasets /PartType0/<name>.
 *
 * The reference appends the datasets through libhdf5; here the file is read
 * with Hdf5Reader and written anew with Hdf5Writer (same groups, attributes
 * and datasets, plus the lines), then moved over the original.
 */
#ifndef CMI_EMISSIVITYCALCULATIONSIMULATION_HPP
#define CMI_EMISSIVITYCALCULATIONSIMULATION_HPP

#include "GpuIonizationSimulation.hpp"
#include "Hdf5Reader.hpp"

#include <cstdio>
#include <iostream>

namespace cmi {

class EmissivityCalculationSimulation {
  static void check(int rc, const char *what) {
    if (rc != CMI_GPU_OK)
      throw std::runtime_error(std::string(what) + ": " +
                               cmi_gpu_last_error());
  }

public:
  /* EmissivityCalculationSimulation::do_simulation, :58-299 */
  static int do_simulation(const std::string &parameterfile_name,
                           const std::string &input_file_name, int device,
                           bool write_output, bool verbose) {
    auto status = [verbose](const std::string &text) {
      if (verbose)
        std::cout << text << std::endl;
    };
    ParameterFile params(parameterfile_name);
    std::vector<int32_t> lines;
    for (int32_t i = 0; i < CMI_GPU_NUMBER_OF_EMISSIONLINES; ++i)
      if (params.get_bool(
              std::string("EmissivityValues:") +
                  GpuIonizationSimulation::emission_line_name(i),
              false))
        lines.push_back(i);
    if (write_output) {
      std::ofstream pfile(parameterfile_name + ".used-values");
      params.print_contents(pfile);
    }
    if (input_file_name.empty())
      throw ParameterError("No input file name provided (--file)!");
    status("Reading file \"" + input_file_name + "\"...");

    Hdf5Reader file(input_file_name);
    /* :101-117 */
    ParameterFile simulation_parameters;
    for (const auto &kv : file.open("/Parameters").attributes)
      simulation_parameters.add_value(kv.first,
                                      Hdf5Reader::as_string(kv.second));
    /* :123-147 */
    double unit_number_density_in_SI = 1., unit_temperature_in_SI = 1.;
    if (file.exists("/Units")) {
      const Hdf5Reader::Object units = file.open("/Units");
      const double unit_length_in_SI =
          0.01 * Hdf5Reader::as_doubles(
                     units.attributes.at("Unit length in cgs (U_L)"))
                     .at(0);
      unit_number_density_in_SI =
          1. / unit_length_in_SI / unit_length_in_SI / unit_length_in_SI;
      unit_temperature_in_SI =
          Hdf5Reader::as_doubles(
              units.attributes.at("Unit temperature in cgs (U_T)"))
              .at(0);
    }
    /* :153-165: old parameter files name the abundances directly */
    double abundances[6];
    if (simulation_parameters.has_value("Abundances:helium")) {
      static const char *names[6] = {"helium", "carbon",  "nitrogen",
                                     "oxygen", "neon",    "sulphur"};
      static const double defaults[6] = {0.1,    2.2e-4, 4.e-5,
                                         3.3e-4, 5.e-5,  9.e-6};
      for (int i = 0; i < 6; ++i)
        abundances[i] = simulation_parameters.get_double(
            std::string("Abundances:") + names[i], defaults[i]);
    } else {
      const Abundances model(simulation_parameters);
      for (int i = 0; i < 6; ++i)
        abundances[i] = model.value[i];
    }

    /* :209-229 */
    std::vector<double> number_density =
        file.read_doubles("/PartType0/NumberDensity");
    std::vector<double> temperature =
        file.read_doubles("/PartType0/Temperature");
    const size_t size = number_density.size();
    if (temperature.size() != size)
      throw ParameterError("snapshot with " + std::to_string(size) +
                           " densities and " +
                           std::to_string(temperature.size()) +
                           " temperatures");
    const std::array<long long, 3> ncell =
        simulation_parameters.get_integer_vector("DensityGrid:number of cells",
                                                 {-1, -1, -1});
    if ((long long)size != ncell[0] * ncell[1] * ncell[2])
      throw ParameterError(
          "the snapshot does not hold DensityGrid:number of cells cells");
    std::vector<double> fractions((size_t)NUMBER_OF_IONNAMES * size);
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion) {
      const std::string name =
          std::string("/PartType0/NeutralFraction") + ion_name(ion);
      if (!file.exists(name))
        throw ParameterError(std::string("Missing ionic fractions for \"") +
                             ion_name(ion) + "\"!");
      const std::vector<double> x = file.read_doubles(name);
      if (x.size() != size)
        throw ParameterError("dataset " + name + " has the wrong size");
      std::copy(x.begin(), x.end(), fractions.begin() + (size_t)ion * size);
    }
    for (size_t i = 0; i < size; ++i) {
      number_density[i] *= unit_number_density_in_SI;
      temperature[i] *= unit_temperature_in_SI;
    }

    status("Starting emissivity calculation...");
    std::vector<double> values(lines.size() * size);
    if (!lines.empty()) {
      /* the cells in the file's order on a device grid of the same shape
       * (every cell on its own: the shape only has to hold them) */
      cmi_gpu_config config = {};
      for (int a = 0; a < 3; ++a) {
        config.anchor[a] = 0.;
        config.sides[a] = 1.;
        config.ncell[a] = (int32_t)ncell[a];
      }
      config.device = device;
      cmi_gpu_engine *engine = nullptr;
      check(cmi_gpu_create(&config, &engine), "cmi_gpu_create");
      int rc = cmi_gpu_set_abundances(engine, abundances);
      if (rc == CMI_GPU_OK)
        rc = cmi_gpu_upload_cells(engine, number_density.data(),
                                  temperature.data(), fractions.data());
      if (rc == CMI_GPU_OK)
        rc = cmi_gpu_compute_emissivities(engine, (int32_t)lines.size(),
                                          lines.data(), 0, (int64_t)size,
                                          values.data());
      const std::string message = rc ? cmi_gpu_last_error() : "";
      cmi_gpu_destroy(engine);
      if (rc)
        throw std::runtime_error("emissivity calculation: " + message);
    }
    status("Finished emissivity calculation.");

    /* :181-193,258-263: the lines as datasets of /PartType0 (existing ones
     * are overwritten). Everything else in the file is carried over. */
    Hdf5Writer out;
    std::vector<std::vector<double>> kept; /* alive until out.write() */
    const Hdf5Reader::Object root = file.open("/");
    size_t ndatasets = 0;
    for (const auto &g : root.members)
      ndatasets += file.object(g.second).members.size();
    kept.reserve(ndatasets);
    for (const auto &g : root.members) {
      const Hdf5Reader::Object group = file.object(g.second);
      if (!group.is_group)
        throw ParameterError("\"/" + g.first + "\" is not a group: this file "
                             "cannot be rewritten with the lines added");
      out.create_group(g.first);
      for (const auto &a : group.attributes)
        out.attribute_raw(g.first, a.first, a.second.type.cls,
                          a.second.type.size, a.second.type.is_signed,
                          a.second.dims, a.second.data);
      for (const auto &d : group.members) {
        bool replaced = false;
        if (g.first == "PartType0")
          for (int32_t line : lines)
            replaced = replaced ||
                       d.first ==
                           GpuIonizationSimulation::emission_line_name(line);
        if (replaced) {
          std::cout << "Warning: dataset \"" << d.first
                    << "\" already exists! Values will be overwritten!"
                    << std::endl;
          continue;
        }
        const Hdf5Reader::Object ds = file.object(d.second);
        if (ds.is_group || ds.type.cls != 1 || ds.type.size != 8)
          throw ParameterError("\"/" + g.first + "/" + d.first +
                               "\" is not a dataset of doubles: this file "
                               "cannot be rewritten with the lines added");
        const std::vector<uint8_t> raw = file.raw(ds);
        kept.emplace_back(raw.size() / 8);
        std::memcpy(kept.back().data(), raw.data(), raw.size());
        const std::vector<double> *v = &kept.back();
        out.dataset(g.first, d.first, ds.dims, [v](std::ostream &os) {
          os.write(reinterpret_cast<const char *>(v->data()), 8 * v->size());
        });
      }
    }
    for (size_t k = 0; k < lines.size(); ++k) {
      const double *v = values.data() + k * size;
      out.dataset("PartType0",
                  GpuIonizationSimulation::emission_line_name(lines[k]),
                  {size}, [v, size](std::ostream &os) {
                    os.write(reinterpret_cast<const char *>(v), 8 * size);
                  });
    }
    const std::string temporary = input_file_name + ".tmp";
    out.write(temporary);
    if (std::rename(temporary.c_str(), input_file_name.c_str()) != 0)
      throw std::runtime_error("could not replace \"" + input_file_name +
                               "\"");
    status("Closed file.");
    return 0;
  }
};

} // namespace cmi

#endif
