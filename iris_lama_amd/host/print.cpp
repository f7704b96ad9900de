// print.cpp -- lama::print / lama::format (include/lama/print.h of the reference: src/print.cpp).
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "lama/print.h"

namespace lama {

void print(const char* format, ...)
{
    va_list args;
    va_start(args, format);
    std::vprintf(format, args);
    va_end(args);
}

std::string format(const char* format, ...)
{
    va_list args, again;
    va_start(args, format);
    va_copy(again, args);
    const int n = std::vsnprintf(nullptr, 0, format, args);
    va_end(args);
    std::string out;
    if (n > 0) {
        std::vector<char> buf((size_t)n + 1);
        std::vsnprintf(buf.data(), buf.size(), format, again);
        out.assign(buf.data(), (size_t)n);
    }
    va_end(again);
    return out;
}

} // namespace lama
