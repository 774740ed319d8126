// Compile-and-run check of the host-side mirrors of the reference's types (include/feature.h, view.hpp, states.h):
// code written against the reference's member names must build against host/include/slam_host.hpp.
//   typecheck            View / subtract / toMat and Feature accessors (no GPU needed)
//   typecheck gpu        also one frame through States and the `features`, rf, nrf mirror
#include "include/slam_host.hpp"

using namespace slamhost;

int main(int argc, char** argv)
{
    // measurementModel-style use of View (src/slam/main.cpp:351-386) and the innovation (:223-225)
    View meas, hmu, err;
    meas.altitude = -1.34; hmu.altitude = -1.30;
    meas.features.push_back(Vfeat(Point2d(0.1, 0.2), Point2d(0.3, 0.4), Point2d(0.5, 0.6)));
    hmu.features.push_back(Vfeat(Vec3d(0.11, 0.19, 0), Vec3d(0.29, 0.41, 0), Vec3d(0.52, 0.57, 0)));
    meas.features.push_back(Vfeat(Point2d(1.1, 1.2), Point2d(1.3, 1.4)));                  // no reflection
    hmu.features.push_back(Vfeat(Vec3d(1.0, 1.0, 0), Vec3d(1.0, 1.0, 0)));
    subtract(meas, hmu, err);
    std::vector<double> e;
    err.toMat(e);
    printf("rows %zu:", e.size());
    for (double v : e) printf(" %.17g", v);
    printf("\n");
    std::vector<int> rfl;                                                                   // main.cpp:211-218
    for (size_t i = 0; i < meas.features.size(); ++i) rfl.push_back(meas.features[i].reflection == NONREF ? 0 : 1);
    printf("refl flags %d %d\n", rfl[0], rfl[1]);

    Feature f;
    f.setID(7).set_body(Vec3d(0.1, 0.2, 0.01));
    f.set_initial_anchor(Vec3d(1, 2, 3));
    f.set_initial_quaternion(Quaternion(0, 0, 0, 1));
    f.set_initial_pib(Point2d(0.1, 0.2));
    f.set_isRef(true);
    printf("feature %d body %g %g %g anchor %g isRef %d noMatch %d\n", f.getID(), f.get_body_position()[0],
           f.get_body_position()[1], f.get_body_position()[2], f.get_initial_anchor()[2], (int)f.initial.isRef, f.get_noMatch());

    if (argc > 1 && !strcmp(argv[1], "gpu")) {
        States mu(8);
        mu.X = Vec3d(0.5, -0.25, 1.34);
        mu.pushX();
        Sensors sense;
        sense.quat.set_value(1.0, Quaternion(0, 0, 0.1, 0.99));
        ImageSensor* img = nullptr;
        (void)img;
        std::vector<slamk_match> m(3);
        for (int i = 0; i < 3; ++i) { m[i].id = 10 + i; m[i].sx = 0.1 * i; m[i].sy = -0.05; m[i].rx = i == 1 ? 0.1 : 0; m[i].ry = i == 1 ? 0.3 : 0; }
        const Quaternion q = sense.quat.get_value();
        if (slamk_frame(mu.h, 0, m.data(), 3, q.coord.v)) return 2;
        mu.pull();
        mu.pull_features();
        printf("nf %d rf %d nrf %d rows %d\n", mu.getNumFeatures(), mu.rf, mu.nrf, mu.getRows());
        for (cFiter ft = mu.features.begin(); ft != mu.features.end(); ++ft)          // main.cpp:358-360 iteration
            printf("id %d pib %g %g %g anchor %g %g %g q0w %g pib0 %g isRef %d\n", ft->getID(), ft->get_body_position()[0],
                   ft->get_body_position()[1], ft->get_body_position()[2], ft->initial.anchor[0], ft->initial.anchor[1],
                   ft->initial.anchor[2], ft->get_initial_quaternion().coord[3], ft->initial.pib.x, (int)ft->initial.isRef);
    }
    return 0;
}
