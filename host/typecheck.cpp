// Compile-and-run check of the host-side mirrors of the reference's types (include/feature.h, view.hpp, states.h):
// code written against the reference's member names must build against host/include/slam_host.hpp.
//   typecheck            View / subtract / toMat, Feature accessors and prediction methods, States value arithmetic
//                        (no GPU needed)
//   typecheck gpu        also one frame through States and the `features`, rf, nrf mirror, and one iteration with the
//                        host mirrors of States::dynamics and Feature::*Hat beside the device
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

    // States as a plain value: the correction K e (src/ARC/states.cpp:24-37), mu += it (:17-22, 69-86), scaling (:6-14:
    // b is not scaled) and the increment of the motion model (:222-263)
    {
        const std::vector<double> kx = {1, 2, 3, 4, 5, 6, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 9, 9, 9};
        States d(kx);
        States mu;
        mu.setX(Vec3d(10, 20, 30)).setV(Vec3d(1, 0.5, 0.25)).setb(Vec3d(0.5, 0.5, 0.5));
        mu.addFeature(Feature(Vec3d(0.1, 0.2, 0.05), 7)).addFeature(Feature(Vec3d(-0.1, 0.3, 0.02), 8));
        mu += d;
        printf("value add: nf %d X %.17g V %.17g f1 %.17g b %.17g id %d\n", d.getNumFeatures(), mu.X[2], mu.V[0],
               mu.features[1].get_body_position()[2], mu.b[0], mu.features[1].getID());
        States half = mu * 0.5;
        printf("value scale: X %.17g f0 %.17g b %.17g\n", half.X[0], half.features[0].get_body_position()[1], half.b[1]);
        Sensors sense;
        sense.quat.set_value(1.0, Quaternion(0, 0, 0, 1));
        sense.ang.set_value(1.0, Vec3d(0.1, 0.2, 0.3));
        sense.acc.set_value(1.0, Vec3d(1, 2, 3));
        const States inc = mu.dynamics(sense, 0.5);
        const Vec3d fp = inc.features[0].get_body_position();
        printf("value dynamics: X %.17g V %.17g f0 %.17g %.17g %.17g b %.17g\n", inc.X[1], inc.V[2], fp[0], fp[1], fp[2], inc.b[0]);
    }
    // Feature::*Hat (src/ARC/feature.cpp:110-198) on a case with identity attitudes
    {
        Feature g(Vec3d(0.1, 0.2, 0.05), 3);
        g.set_initial_anchor(Vec3d(1, 2, 3));
        g.set_initial_quaternion(Quaternion(0, 0, 0, 1));
        const Quaternion qbw(0, 0, 0, 1);
        const Vec3d xi = g.xibHat(), p0 = g.pib0Hat(Vec3d(1.5, 2.5, 3.5), qbw), pp = g.ppbHat(Vec3d(0, 0, 1.34), qbw);
        printf("hat: xib %.17g %.17g %.17g pib0 %.17g %.17g ppb %.17g %.17g\n", xi[0], xi[1], xi[2], p0[0], p0[1], pp[0], pp[1]);
    }

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

        // One iteration with an image, the host mirrors beside the device (src/slam/main.cpp:167-225):
        //   mu += mu.dynamics(sense, dt) on a value copy against slamk_predict,
        //   meas - hmu from Feature::pib0Hat(old_pos) / ppbHat(X) and View::subtract / toMat against the device innovation.
        sense.ang.set_value(1.0, Vec3d(0.01, -0.02, 0.015));
        sense.acc.set_value(1.0, Vec3d(0.2, -0.1, 0.05));
        mu.V = Vec3d(1.0, 0.1, -0.05);
        mu.pushV();
        const double dt = 1.0 / 30;
        const Vec3d old_pos = mu.X;                                                   // main.cpp:168
        States hv = mu;                                                               // value copy: no handle
        hv += hv.dynamics(sense, dt);                                                 // :193-194
        mu.predict(sense, dt);
        mu.pull();
        mu.pull_features();
        double ddyn = 0;
        for (int c = 0; c < 3; ++c) ddyn = std::fmax(ddyn, std::fmax(std::fabs(hv.X[c] - mu.X[c]), std::fabs(hv.V[c] - mu.V[c])));
        for (size_t i = 0; i < mu.features.size(); ++i)
            for (int c = 0; c < 3; ++c)
                ddyn = std::fmax(ddyn, std::fabs(hv.features[i].get_body_position()[c] - mu.features[i].get_body_position()[c]));
        printf("dynamics mirror: max abs deviation %.3g\n", ddyn);
        View vm, vh, ve;
        vm.altitude = -1.34;                                                          // :177
        vh.altitude = -mu.X[2];                                                       // :352
        for (size_t i = 0; i < mu.features.size(); ++i) {
            const Feature& ft = mu.features[i];
            const Vec3d bp = ft.get_body_position();
            const Point2d src(m[i].sx, m[i].sy), refl(m[i].rx, m[i].ry);
            if (ft.initial.isRef) {
                vm.features.push_back(Vfeat(src, ft.initial.pib, refl));
                vh.features.push_back(Vfeat(Vec3d(bp[0], bp[1], 0), ft.pib0Hat(old_pos, q), ft.ppbHat(mu.X, q)));
            } else {
                vm.features.push_back(Vfeat(src, ft.initial.pib));
                vh.features.push_back(Vfeat(Vec3d(bp[0], bp[1], 0), ft.pib0Hat(old_pos, q)));
            }
        }
        subtract(vm, vh, ve);
        std::vector<double> eh;
        ve.toMat(eh);
        mu.update(sense);
        std::vector<double> ed(slamk_num_rows(mu.h, 0));
        if (slamk_debug_get(mu.h, 0, nullptr, nullptr, nullptr, ed.data())) return 3;
        double dinn = eh.size() == ed.size() ? 0 : 1e300;
        for (size_t r = 0; r < eh.size() && r < ed.size(); ++r) dinn = std::fmax(dinn, std::fabs(eh[r] - ed[r]));
        printf("innovation mirror: rows %zu max abs deviation %.3g\n", ed.size(), dinn);
    }
    return 0;
}
