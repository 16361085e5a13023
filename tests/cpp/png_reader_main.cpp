// Stand-alone driver of the native host layer's PNG reader (easysfm_amd/host/esfm_png.hpp) for tests/test_host_logic.py: built
// with -fsanitize=address,undefined on the CPU so that malformed and sub-byte files are checked for out-of-bounds accesses.
//   png_reader_main in.png [out.raw]   -> prints "rows cols", writes the BGR bytes; exit 3 + "error: ..." when the file is refused
#include "../../easysfm_amd/host/esfm_png.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    int rows = 0, cols = 0;
    std::vector<uint8_t> bgr;
    const std::string e = p3dv::png::read_bgr(argv[1], rows, cols, bgr);
    if (!e.empty()) { std::printf("error: %s\n", e.c_str()); return 3; }
    std::printf("%d %d\n", rows, cols);
    if (argc > 2) {
        FILE *f = std::fopen(argv[2], "wb");
        if (!f) return 2;
        std::fwrite(bgr.data(), 1, bgr.size(), f);
        std::fclose(f);
    }
    return 0;
}
