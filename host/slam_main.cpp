// `slam <featurefile> [ignored...] < sensors` -- the estimator's text front end on the MI355X path.
// Same command line, stdin/feature-file formats, stdout line and exit behaviour as the reference
// binary (src/slam/main.cpp:100-238); the numerical work goes through the C ABI (include/slamk.h).
#include "include/slam_host.hpp"

using namespace slamhost;

static void help(const char* str) { fprintf(stderr, "Usage: %s [measurements] < sensors\n", str); }

int main(int argc, char** argv)
{
    if (argc < 2 || !strcmp(argv[1], "-h")) {   // main.cpp:109-112
        help(argv[0]);
        exit(EXIT_FAILURE);
    }
    const char* cap = getenv("SLAM_MAX_FEATURES");
    States mu(cap ? atoi(cap) : 256);
    Sensors sense;
    ImageSensor imgsense(argv[1], false);

    int u = sense.update();                      // :131
    if (u & UPDATE_INIT) {                       // :132-137 (x and y swapped)
        Vec3d pos = sense.init.get_value();
        mu.X = Vec3d(pos[1], pos[0], pos[2]);
        mu.pushX();
    }
    if (!(u & UPDATE_IMG)) {                     // :138-141
        fprintf(stderr, "No image measurement at first time step.\n");
        exit(EXIT_FAILURE);
    }
    imgsense.update();                           // :142
    mu.update_features(imgsense, sense);         // :143 (old_pos = X is taken inside the frame call, :144)

    double dt = 0;
    while (u != -1) {                            // :150
        u = sense.update();                      // :157
        if (u & UPDATE_ACC) dt = sense.acc.get_dt();          // :158-166
        else if (u & UPDATE_ANG) dt = sense.ang.get_dt();
        else if (u & UPDATE_QUAT) dt = sense.quat.get_dt();
        else if (u & UPDATE_ALT) dt = sense.alt.get_dt();
        else if (u == 0) {
            fprintf(stderr, "No update, so what are we doing here?\n");
            exit(EXIT_FAILURE);
        }
        if (u & UPDATE_IMG) {                    // :167-175
            imgsense.update();
            mu.update_features(imgsense, sense);
        }
        // :177 the altitude override lives in slamk_params.use_alt_override (default on)
        if (u & UPDATE_POS) {                    // :180-185
            Vec3d pos = sense.pos.get_value();
            mu.X = Vec3d(pos[1], pos[0], pos[2]);
            mu.pushX();
        }
        if (u & UPDATE_VEL) {                    // :186-191
            Matx33d Rw2b = sense.quat.get_value().rotation().t();
            mu.V = Rw2b * sense.vel.get_value();
            mu.pushV();
        }
        mu.predict(sense, dt);                   // :192-194, :204-206
        mu.pull();
        printf("%0.5f,%0.5f\n", mu.X[1], mu.X[0]);            // :207
        if (u & UPDATE_IMG) mu.update(sense);    // :198-202, :209-233
    }
    return 0;
}
