/* Test support: the values testParameterFile.cpp asks the reference's
 * ParameterFile for (test/testParameterFile.cpp:78-150), read with the host's
 * parser from the file given on the command line, as one JSON object. */
#include "ParameterFile.hpp"

#include <cstdio>
#include <iostream>

int main(int argc, char **argv) {
  if (argc != 2)
    return 2;
  try {
    cmi::ParameterFile params(argv[1]);
    std::printf("{");
    for (int i = 1; i <= 5; ++i)
      std::printf("\"test_integer%d\": %lld, ", i,
                  params.get_integer("test_integer" + std::to_string(i), -1));
    std::printf("\"test_float\": %.17g, ", params.get_double("test_float", -1.));
    std::printf("\"test_unit\": %.17g, ",
                params.get_physical_value(cmi::QUANTITY_LENGTH, "test_unit",
                                          "-1. m"));
    for (int i = 1; i <= 8; ++i)
      std::printf("\"test_bool%d\": %s, ", i,
                  params.get_bool("test_bool" + std::to_string(i), i > 4)
                      ? "true"
                      : "false");
    std::printf("\"test_string\": \"%s\", ",
                params.get_string("test_string", "").c_str());
    std::printf("\"group_member\": %lld, ",
                params.get_integer("test_group:test_group_member", -1));
    std::printf("\"comments_value\": \"%s\", ",
                params.get_string("test_comments_group:test_comments_value", "")
                    .c_str());
    const auto d = params.get_physical_vector(
        cmi::QUANTITY_LENGTH, "test_coordinatevector_unit", "");
    std::printf("\"vector_unit\": [%.17g, %.17g, %.17g], ", d[0], d[1], d[2]);
    const auto iv =
        params.get_integer_vector("test_coordinatevector_int", {-1, -1, -1});
    std::printf("\"vector_int\": [%lld, %lld, %lld], ", iv[0], iv[1], iv[2]);
    const auto bv = params.get_bool_vector("test_coordinatevector_bool",
                                           {true, false, false});
    std::printf("\"vector_bool\": [%s, %s, %s], ", bv[0] ? "true" : "false",
                bv[1] ? "true" : "false", bv[2] ? "true" : "false");
    std::printf(
        "\"group_group_member\": %lld, ",
        params.get_integer(
            "test_group2:test_group_group:test_group_group_member", -1));
    /* default values */
    std::printf("\"not_in_file1\": %lld, ",
                params.get_integer("not_in_file1", 42));
    std::printf("\"not_in_file2\": %.17g, ",
                params.get_double("not_in_file2", 3.14));
    std::printf("\"unit_not_in_file\": %.17g, ",
                params.get_physical_value(cmi::QUANTITY_LENGTH,
                                          "unit_not_in_file", "1. pc"));
    std::printf("\"not_in\": \"%s\", ",
                params.get_string("not_in", "file?").c_str());
    std::printf("\"not_in_file3\": %s}\n",
                params.get_bool("not_in_file3", true) ? "true" : "false");
  } catch (const std::exception &e) {
    std::cerr << e.what() << "\n";
    return 1;
  }
  return 0;
}
