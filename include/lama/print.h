// lama/print.h -- printf-style helpers with the reference's interface (include/lama/print.h:40-46).
#pragma once
#include <string>

namespace lama {

// formatted text to stdout
void print(const char* format, ...) __attribute__((__format__(__printf__, 1, 2)));
// formatted text as a string
std::string format(const char* format, ...) __attribute__((__format__(__printf__, 1, 2)));

} // namespace lama
