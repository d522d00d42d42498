// TEST TOOLING: a tiny msdfgen *client* written against the reference's public headers exactly like the README example
// (README.md:114-142): read a shape description, normalize, colour, generateMSDF / generateMTSDF / generateSDF, dump raw floats.
// It is linked against msdfgen_amd's C++ shim (which provides msdfgen::generate*) plus the reference's remaining objects
// (shape description parser, Shape, edge colouring ...) -- never against the reference's own msdfgen.o / msdf-error-correction.o.
//   usage: shim_check <shapedesc-file> <out.bin> <mode 1|2|3|4> <w> <h> <scale> <tx> <ty> <range> [ydown]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "msdfgen.h"

using namespace msdfgen;

int main(int argc, char **argv) {
    if (argc < 10)
        return 2;
    FILE *f = fopen(argv[1], "r");
    Shape shape;
    if (!f || !readShapeDescription(f, shape))
        return 3;
    fclose(f);
    shape.normalize();
    edgeColoringSimple(shape, 3.0);
    const int mode = atoi(argv[3]), w = atoi(argv[4]), h = atoi(argv[5]);
    const double scale = atof(argv[6]), tx = atof(argv[7]), ty = atof(argv[8]), range = atof(argv[9]);
    const YAxisOrientation yo = argc > 10 && atoi(argv[10]) ? Y_DOWNWARD : Y_UPWARD;
    const int N = mode <= 2 ? 1 : mode;
    std::vector<float> px((size_t) w*h*N);
    SDFTransformation t(Projection(scale, Vector2(tx, ty)), Range(range));
    try {
        switch (mode) {
            case 1: generateSDF(BitmapSection<float, 1>(px.data(), w, h, yo), shape, t); break;
            case 2: generatePSDF(BitmapSection<float, 1>(px.data(), w, h, yo), shape, t); break;
            case 3: generateMSDF(BitmapSection<float, 3>(px.data(), w, h, yo), shape, t); break;
            default: generateMTSDF(BitmapSection<float, 4>(px.data(), w, h, yo), shape, Projection(scale, Vector2(tx, ty)), Range(range)); break;
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 4;
    }
    f = fopen(argv[2], "wb");
    fwrite(px.data(), sizeof(float), px.size(), f);
    fclose(f);
    return 0;
}
