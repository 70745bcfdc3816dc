/*
 * cmi_gpu_main.cpp - the `cmi-gpu` executable: runs an ionization simulation
 * described by a CMacIonize ".param" file on the MI355X engine.
 *
 * Command line of the reference's driver for this path
 * (src/CMacIonize.cpp:113-179,306-377): --params/-p, --threads/-t,
 * --every-iteration-output/-e, --output-statistics/-s, --dry-run/-n,
 * --verbose/-v, and the emission mode --emission/-m with --file/-f
 * (EmissivityCalculationSimulation.hpp), and --task-based: the run takes its
 * control parameters from the TaskBasedIonizationSimulation: block of the
 * file (src/CMacIonize.cpp:335-345, src/TaskBasedIonizationSimulation.cpp:
 * 190-260) instead of IonizationSimulation:. Flags that select other code paths
 * of the reference (--rhd, --dusty-radiative-transfer, --task-based-rhd) are
 * rejected.
 * New: --device N (HIP device ordinal), --describe (print the lowered plugin
 * descriptors as JSON; with --dry-run no GPU is needed), --blocks BX,BY,BZ
 * (domain decomposition: one engine per block of the grid, the counterpart of
 * the reference's DensitySubGridCreator:number of subgrids) and
 * --devices D0,D1,... (the blocks' devices, round-robin; default --device),
 * --copies K (K engines for every block that contains a source - the
 * reference's TaskBasedIonizationSimulation:source copy level; default: one
 * per device, 2^level at most).
 */
#include "EmissivityCalculationSimulation.hpp"
#include "GpuIonizationSimulation.hpp"

#include <cstring>
#include <iostream>

using namespace cmi;

static void describe(GpuIonizationSimulation &sim) {
  std::cout.precision(17);
  ParameterFile &p = sim.parameter_file();
  (void)p;
  const DensityGrid &g = sim.grid();
  std::cout << "{\n";
  std::cout << "  \"anchor\": [" << g.box().anchor[0] << ", "
            << g.box().anchor[1] << ", " << g.box().anchor[2] << "],\n";
  std::cout << "  \"sides\": [" << g.box().sides[0] << ", " << g.box().sides[1]
            << ", " << g.box().sides[2] << "],\n";
  std::cout << "  \"periodicity\": [" << g.box().periodicity[0] << ", "
            << g.box().periodicity[1] << ", " << g.box().periodicity[2]
            << "],\n";
  std::cout << "  \"ncell\": [" << g.ncell()[0] << ", " << g.ncell()[1] << ", "
            << g.ncell()[2] << "],\n";
  std::cout << "  \"number_of_iterations\": " << sim.number_of_iterations()
            << ",\n  \"number_of_photons\": " << sim.number_of_photons()
            << ",\n  \"random_seed\": " << sim.random_seed() << ",\n";
  PhotonSourceDistribution *src = sim.sources();
  std::cout << "  \"sources\": [";
  for (photonsourcenumber_t i = 0; src && i < src->get_number_of_sources();
       ++i) {
    const CoordinateVector x = src->get_position(i);
    std::cout << (i ? ", " : "") << "{\"position\": [" << x.x() << ", " << x.y()
              << ", " << x.z() << "], \"weight\": " << src->get_weight(i)
              << "}";
  }
  std::cout << "],\n  \"total_luminosity\": "
            << (src ? src->get_total_luminosity() : 0.) << ",\n";
  if (auto *m = dynamic_cast<MonochromaticPhotonSourceSpectrum *>(
          sim.spectrum()))
    std::cout << "  \"spectrum\": {\"type\": \"Monochromatic\", \"frequency\": "
              << m->get_frequency() << "},\n";
  else if (auto *pl =
               dynamic_cast<PlanckPhotonSourceSpectrum *>(sim.spectrum()))
    std::cout << "  \"spectrum\": {\"type\": \"Planck\", \"temperature\": "
              << pl->get_temperature() << "},\n";
  else if (sim.spectrum()) {
    /* known through its virtuals alone: what the generic lowering made of it */
    const SpectrumTable table = tabulate_spectrum(*sim.spectrum());
    std::cout << "  \"spectrum\": {\"type\": " << sim.spectrum()->describe()
              << ", \"lowering\": \"" << table.method
              << "\", \"samples\": " << table.frequency.size()
              << ", \"minimum_frequency\": " << table.frequency.front()
              << ", \"maximum_frequency\": " << table.frequency.back()
              << "},\n";
  }
  std::cout << "  \"cross_sections\": ";
  if (dynamic_cast<FixedValueCrossSections *>(sim.cross_sections())) {
    std::cout << "[";
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      std::cout << (ion ? ", " : "")
                << sim.cross_sections()->get_cross_section(ion, 0.);
    std::cout << "],\n";
  } else if (dynamic_cast<VernerCrossSections *>(sim.cross_sections())) {
    std::cout << "\"Verner\",\n";
  } else {
    std::cout << "{\"type\": " << sim.cross_sections()->describe()
              << ", \"samples\": " << sim.cross_sections()->tabulate().x.size()
              << "},\n";
  }
  std::cout << "  \"recombination_rates\": ";
  if (dynamic_cast<FixedValueRecombinationRates *>(sim.recombination_rates())) {
    std::cout << "[";
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      std::cout << (ion ? ", " : "")
                << sim.recombination_rates()->get_recombination_rate(ion, 0.);
    std::cout << "],\n";
  } else if (dynamic_cast<VernerRecombinationRates *>(
                 sim.recombination_rates())) {
    std::cout << "\"Verner\",\n";
  } else {
    std::cout << "{\"type\": " << sim.recombination_rates()->describe()
              << ", \"samples\": "
              << sim.recombination_rates()->tabulate().x.size() << "},\n";
  }
  std::cout << "  \"abundances\": [";
  for (int i = 0; i < 6; ++i)
    std::cout << (i ? ", " : "") << sim.abundances().value[i];
  std::cout << "],\n  \"reemission\": {\"type\": " << sim.reemission().type
            << ", \"probability\": " << sim.reemission().probability
            << ", \"frequency\": " << sim.reemission().frequency << "},\n";
  const cmi_gpu_temperature_params &t = sim.temperature_params();
  std::cout << "  \"temperature\": {\"do\": " << t.do_temperature_calculation
            << ", \"min_iterations\": " << t.minimum_number_of_iterations
            << ", \"epsilon\": " << t.epsilon_convergence
            << ", \"max_iterations\": " << t.maximum_number_of_iterations
            << ", \"pah\": " << t.pah_heating_factor
            << ", \"cr_factor\": " << t.cosmic_ray_heating_factor
            << ", \"cr_limit\": " << t.cosmic_ray_heating_limit
            << ", \"cr_scale\": " << t.cosmic_ray_heating_scale_length
            << ", \"T_min_ionized\": " << t.minimum_ionized_temperature
            << "}\n}\n";
}

int main(int argc, char **argv) {
  std::string params;
  int threads = 1, device = 0;
  std::array<int, 3> blocks = {1, 1, 1};
  std::vector<int> devices;
  int copies = 0;
  auto int_list = [](const std::string &text) {
    std::vector<int> values;
    std::stringstream stream(text);
    std::string item;
    while (std::getline(stream, item, ','))
      values.push_back(std::atoi(item.c_str()));
    return values;
  };
  bool dry_snapshot = false;
  bool emission = false;
  std::string input_file;
  bool every_iteration = false, statistics = false, dry_run = false,
       verbose = false, do_describe = false, task_based = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto need = [&](const char *name) -> std::string {
      if (i + 1 >= argc) {
        std::cerr << "Missing argument for " << name << "\n";
        exit(1);
      }
      return argv[++i];
    };
    if (a == "--params" || a == "-p")
      params = need("--params");
    else if (a == "--threads" || a == "-t")
      threads = std::atoi(need("--threads").c_str());
    else if (a == "--device")
      device = std::atoi(need("--device").c_str());
    else if (a == "--blocks") {
      const std::vector<int> b = int_list(need("--blocks"));
      if (b.size() != 3 || b[0] < 1 || b[1] < 1 || b[2] < 1) {
        std::cerr << "--blocks needs three positive integers BX,BY,BZ\n";
        return 1;
      }
      blocks = {b[0], b[1], b[2]};
    } else if (a == "--devices")
      devices = int_list(need("--devices"));
    else if (a == "--copies")
      copies = std::atoi(need("--copies").c_str());
    else if (a == "--emission" || a == "-m")
      emission = true;
    else if (a == "--file" || a == "-f")
      input_file = need("--file");
    else if (a == "--every-iteration-output" || a == "-e")
      every_iteration = true;
    else if (a == "--output-statistics" || a == "-s")
      statistics = true;
    else if (a == "--dry-run" || a == "-n")
      dry_run = true;
    else if (a == "--dry-run-snapshot") {
      dry_run = true;
      dry_snapshot = true;
    }
    else if (a == "--verbose" || a == "-v")
      verbose = true;
    else if (a == "--describe")
      do_describe = true;
    else if (a == "--task-based")
      task_based = true; /* src/CMacIonize.cpp: TaskBasedIonizationSimulation */
    else if (a == "--dirty" || a == "-d" || a == "--no-initial-output")
      ; /* accepted, no effect */
    else {
      std::cerr << "Unknown or unsupported option: " << a << "\n"
                << "usage: cmi-gpu --params FILE [--threads N] [--device N] "
                   "[--blocks BX,BY,BZ] [--devices D0,D1,...] [--copies K] "
                   "[--task-based] [--every-iteration-output] [--output-statistics] "
                   "[--dry-run] [--dry-run-snapshot] [--describe] [--verbose]\n"
                   "       cmi-gpu --emission --params FILE --file "
                   "SNAPSHOT.hdf5 [--device N]\n";
      return 1;
    }
  }
  if (params.empty()) {
    std::cerr << "Required option --params missing\n";
    return 1;
  }
  if (emission) {
    /* src/CMacIonize.cpp: the emission mode */
    try {
      return EmissivityCalculationSimulation::do_simulation(
          params, input_file, device, !dry_run, true);
    } catch (std::exception &e) {
      std::cerr << "Error: " << e.what() << std::endl;
      return 1;
    }
  }
  try {
    GpuIonizationSimulation simulation(!dry_run || dry_snapshot,
                                       every_iteration, statistics, threads,
                                       params, device,
                                       verbose || !do_describe, !dry_run,
                                       blocks, devices, copies, task_based);
    if (do_describe)
      describe(simulation);
    if (dry_run && dry_snapshot) {
      /* evaluate the DensityFunction on the host and write snapshot 0 */
      simulation.initialize();
      simulation.write_initial_snapshot();
    }
    if (dry_run) {
      if (!do_describe)
        std::cout << "Dry run successful." << std::endl;
      return 0;
    }
    simulation.initialize();
    simulation.run();
  } catch (std::exception &e) {
    /* the reference aborts (cmac_error); same exit behaviour, no core dump */
    std::cerr << "Error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
