// main.cpp — the `raytracer` binary: main() of the reference (main.rs:22-29) over the C++ host + HIP library.
#include <cstdio>

#include "driver.h"

int main(int argc, char** argv) {
    rth::Cli cli; std::string err;
    if (!rth::parse_cli(argc, argv, cli, err)) {
        bool help = err.rfind("Usage:", 0) == 0;
        fprintf(help ? stdout : stderr, "%s\n", err.c_str());
        return help ? 0 : 2;
    }
    int rc = rth::run(cli, nullptr, err);
    if (rc != 0) fprintf(stderr, "raytracer: %s\n", err.c_str());
    return rc;
}
