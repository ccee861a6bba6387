// inputs_dump.cpp -- prints the first values of the probe's input streams (hex), so that tests/test_pin_kit_cpu.py can hold
// compare.py's Python mirror of pin_inputs.h to the C++ generator without needing Eigen / grid_map / tf2.
#include <cstdio>
#include <cstring>

#include "pin_inputs.h"

int main()
{
    Lcg a(0xE16E0001u);
    for (int k = 0; k < 200; ++k) {
        const float f = a.wide_float();
        uint32_t u;
        std::memcpy(&u, &f, 4);
        std::printf("%08x\n", u);
    }
    Lcg b(0x61D00002u);
    for (int k = 0; k < 200; ++k) {
        const double d = b.unit();
        uint64_t u;
        std::memcpy(&u, &d, 8);
        std::printf("%016llx\n", (unsigned long long)u);
    }
    return 0;
}
