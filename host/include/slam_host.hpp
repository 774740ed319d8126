// Host-side mirror of the reference's types for the estimator's text front end.
//
// The reference's headers (include/Quaternion.hpp, Sensor.hpp, Sensors.hpp, featureIO.h,
// imagesensor.hpp, states.h) embed OpenCV value types; this header keeps their public shape
// (class and member names, argument meaning, error behaviour) on small value types of its own, and
// `States` forwards the numerical work to the C ABI (include/slamk.h) instead of cv::Mat code.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/slamk.h"

namespace slamhost {

struct Vec3d {
    double v[3] = {0, 0, 0};
    Vec3d() {}
    Vec3d(double a, double b, double c) { v[0] = a; v[1] = b; v[2] = c; }
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
    Vec3d& operator+=(const Vec3d& o) { for (int i = 0; i < 3; ++i) v[i] += o.v[i]; return *this; }
    Vec3d& operator-=(const Vec3d& o) { for (int i = 0; i < 3; ++i) v[i] -= o.v[i]; return *this; }
    Vec3d& operator*=(double a) { for (int i = 0; i < 3; ++i) v[i] *= a; return *this; }
    Vec3d operator+(const Vec3d& o) const { return Vec3d(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vec3d operator-(const Vec3d& o) const { return Vec3d(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vec3d operator*(double a) const { return Vec3d(v[0] * a, v[1] * a, v[2] * a); }
};
struct Vec4d {
    double v[4] = {0, 0, 0, 0};
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
};
struct Point2d {
    double x = 0, y = 0;
    Point2d() {}
    Point2d(double a, double b) : x(a), y(b) {}
    bool operator!=(const Point2d& o) const { return x != o.x || y != o.y; }
    bool operator==(const Point2d& o) const { return x == o.x && y == o.y; }
    Point2d operator-(const Point2d& o) const { return Point2d(x - o.x, y - o.y); }
};
struct Matx33d {
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double& operator()(int i, int j) { return m[3 * i + j]; }
    double operator()(int i, int j) const { return m[3 * i + j]; }
    Matx33d t() const
    {
        Matx33d r;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r(i, j) = (*this)(j, i);
        return r;
    }
    Vec3d operator*(const Vec3d& a) const
    {
        Vec3d r;
        for (int i = 0; i < 3; ++i) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += m[3 * i + k] * a[k];
            r[i] = s;
        }
        return r;
    }
    Matx33d operator*(const Matx33d& o) const
    {
        Matx33d r;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += m[3 * i + k] * o.m[3 * k + j];
                r(i, j) = s;
            }
        return r;
    }
};

// include/Quaternion.hpp:11-39, src/ARC/Quaternion.cpp:32-47 (scalar last, not normalised)
class Quaternion {
   public:
    Quaternion() {}
    Quaternion(double q1, double q2, double q3, double q4) { coord[0] = q1; coord[1] = q2; coord[2] = q3; coord[3] = q4; }
    Matx33d rotation() const
    {
        Matx33d m;
        const double* c = coord.v;
        m(0, 0) = pow(c[3], 2) + pow(c[0], 2) - pow(c[1], 2) - pow(c[2], 2);
        m(1, 0) = 2 * (c[0] * c[1] + c[3] * c[2]);
        m(2, 0) = 2 * (c[0] * c[2] - c[3] * c[1]);
        m(0, 1) = 2 * (c[0] * c[1] - c[3] * c[2]);
        m(1, 1) = pow(c[3], 2) - pow(c[0], 2) + pow(c[1], 2) - pow(c[2], 2);
        m(2, 1) = 2 * (c[3] * c[0] + c[1] * c[2]);
        m(0, 2) = 2 * (c[3] * c[1] + c[0] * c[2]);
        m(1, 2) = 2 * (c[1] * c[2] - c[3] * c[0]);
        m(2, 2) = pow(c[3], 2) - pow(c[0], 2) - pow(c[1], 2) + pow(c[2], 2);
        return m;
    }
    Vec4d coord;
};

// include/Sensor.hpp:28-68
template <class T>
class Sensor {
   public:
    Sensor() : value(), timestamp(0), dt(0) {}
    double get_timestamp() const { return timestamp; }
    double get_dt() const { return dt; }
    T get_value() const { return value; }
    void set_value(double ts, T v)
    {
        dt = (timestamp == 0) ? 0 : ts - timestamp;
        timestamp = ts;
        value = v;
    }
    void set_value(T v) { value = v; }

   private:
    T value;
    double timestamp;
    double dt;
};

// include/Sensors.hpp:8-62, src/ARC/Sensors.cpp:4-70
enum {
    UPDATE_ACC = 1, UPDATE_ANG = 1 << 1, UPDATE_QUAT = 1 << 2, UPDATE_IMG = 1 << 3,
    UPDATE_ALT = 1 << 4, UPDATE_INIT = 1 << 5, UPDATE_POS = 1 << 6, UPDATE_VEL = 1 << 7
};

class Sensors {
   public:
    Sensors() : t(0), time(-1), rv(0) { type[0] = vals[0] = 0; }
    // Reads the lines of one time step (stamps closer than 5 ms to the first line) from stdin and
    // returns the update bitmask, or -1 for the group that ends at end of input.
    int update()
    {
        int updated = 0;
        if (time == -1) rv = scanf("%lf,%31[^,\n],%255s", &time, type, vals);
        t = time;
        while (time - t < 0.005) {
            if (rv == EOF) return -1;
            Vec3d v3;
            if (!strcmp(type, "ACC")) { updated |= UPDATE_ACC; sscanf(vals, "%lf,%lf,%lf", &v3[0], &v3[1], &v3[2]); acc.set_value(time, v3); }
            else if (!strcmp(type, "ANG")) { updated |= UPDATE_ANG; sscanf(vals, "%lf,%lf,%lf", &v3[0], &v3[1], &v3[2]); ang.set_value(time, v3); }
            else if (!strcmp(type, "IMG")) { updated |= UPDATE_IMG; img.set_value(time, 0); }
            else if (!strcmp(type, "QUAT")) {
                updated |= UPDATE_QUAT;
                Quaternion q;
                sscanf(vals, "%lf,%lf,%lf,%lf", &q.coord[0], &q.coord[1], &q.coord[2], &q.coord[3]);
                quat.set_value(time, q);
            }
            else if (!strcmp(type, "ALT")) { updated |= UPDATE_ALT; double a = 0; sscanf(vals, "%lf", &a); alt.set_value(time, a); }
            else if (!strcmp(type, "POS")) { updated |= UPDATE_POS; sscanf(vals, "%lf,%lf,%lf", &v3[0], &v3[1], &v3[2]); pos.set_value(time, v3); }
            else if (!strcmp(type, "VEL")) { updated |= UPDATE_VEL; sscanf(vals, "%lf,%lf,%lf", &v3[0], &v3[1], &v3[2]); vel.set_value(time, v3); }
            else if (!strcmp(type, "INIT")) { updated |= UPDATE_INIT; sscanf(vals, "%lf,%lf,%lf", &v3[0], &v3[1], &v3[2]); init.set_value(time, v3); }
            rv = scanf("%lf,%31[^,\n],%255s", &time, type, vals);
        }
        return updated;
    }
    Sensor<Quaternion> quat;
    Sensor<Vec3d> acc, ang, init, vel, pos;
    Sensor<double> alt;
    Sensor<int> img;

   private:
    double t, time;
    int rv;
    char type[32], vals[256];
};

// include/featureIO.h:31-37
struct projection {
    Point2d source, reflection;
    int id = 0;
    bool isRef() const { return reflection != Point2d(0, 0); }
};

// include/imagesensor.hpp:16-28, src/ARC/imagesensor.cpp:29-49, src/ARC/featureIO.cpp:52-59,104-128
class ImageSensor {
   public:
    ImageSensor(const char* fn, bool /*isHex*/)
    {
        fp = fopen(fn, "r");
        if (!fp) { perror("fopen body"); exit(EXIT_FAILURE); }
    }
    ~ImageSensor() { if (fp) fclose(fp); }
    // one frame: lines `id,x,y,rx,ry` up to a blank line (-1) or end of file (-2)
    int update()
    {
        matches.clear();
        char line[1024];
        for (;;) {
            projection pj;
            if (fgets(line, sizeof line, fp) == NULL) return -2;
            int rv = sscanf(line, "%d,%lf,%lf,%lf,%lf", &pj.id, &pj.source.x, &pj.source.y, &pj.reflection.x, &pj.reflection.y);
            if (rv < 0) return rv;
            matches.push_back(pj);
        }
    }
    std::vector<projection> matches;

   private:
    FILE* fp;
};

#define NONREF Point2d(0, 0)         // include/featureIO.h:25

// include/feature.h:15-25
struct frame {
    Vec3d world, body;
};
struct inits {
    Quaternion quaternion;
    double inverse_depth = 0;
    Vec3d anchor;
    Point2d pib;
    bool isRef = false;
};
typedef struct inits Inits;
typedef struct frame Frame;

// include/feature.h:30-222 -- mirror of one tracked feature (States::features is refreshed from the device by
// States::pull_features): same member and accessor names.  The prediction methods (src/ARC/feature.cpp:110-198) are
// what measure_kernel evaluates on the device; the host copies below exist for callers written against the reference's
// class and as an independent check of the device innovation (host/typecheck.cpp).
class Feature {
   public:
    Feature() {}
    Feature(const Vec3d& pos, int n) : ID(n) { position.body = pos; }              // feature.h: Feature(pos, ID)
    inits initial;
    // body-frame point (x forward) of the inverse-depth coordinates (y/x, z/x, 1/x)          feature.cpp:171-177
    Vec3d xibHat() const { return Vec3d(1 / position.body[2], position.body[0] / position.body[2], position.body[1] / position.body[2]); }
    // world offset from the anchor, seen from the initial body frame                          feature.cpp:110-114
    Vec3d fromAnchor(const Vec3d& pos) const { return initial.quaternion.rotation().t() * (pos - initial.anchor); }
    // rotation from the current body frame to the initial one                                 feature.cpp:117-126
    Matx33d rb2b(const Quaternion& qbw) const { return initial.quaternion.rotation().t() * qbw.rotation(); }
    Vec3d xib0Hat(const Vec3d& pos, const Quaternion& qbw) const                    // feature.cpp:178-188
    {
        Vec3d rv = rb2b(qbw) * xibHat();
        rv += fromAnchor(pos);
        return rv;
    }
    Vec3d pib0Hat(const Vec3d& pos, const Quaternion& qbw) const                    // feature.cpp:189-198
    {
        const Vec3d x = xib0Hat(pos, qbw);
        return Vec3d(x[1] / x[0], x[2] / x[0], 0);
    }
    // the feature's mirror image in the water plane z = 0 (n = e_z: S = I - 2 n n^T)         feature.cpp:155-169
    Vec3d xpbHat(const Vec3d& X, const Quaternion& qbw) const
    {
        Vec3d t = qbw.rotation() * xibHat();
        t[2] = -t[2];                        // S * (...)
        t[2] -= 2 * X[2];                    // - 2 n n^T X
        return qbw.rotation().t() * t;
    }
    Vec3d ppbHat(const Vec3d& X, const Quaternion& qbw) const                       // feature.cpp:143-153
    {
        const Vec3d x = xpbHat(X, qbw);
        return Vec3d(x[1] / x[0], x[2] / x[0], 0);
    }
    int getID() const { return ID; }
    Vec3d get_body_position() const { return position.body; }                      // feature.h:101-105
    Vec3d get_initial_anchor() const { return initial.anchor; }                    // :150-154
    Quaternion get_initial_quaternion() const { return initial.quaternion; }       // :170-174
    Point2d get_initial_pib() const { return initial.pib; }
    bool get_isRef() const { return initial.isRef; }
    int get_noMatch() const { return noMatch; }                                    // :196-200
    Feature& setID(int n) { ID = n; return *this; }
    Feature& set_body(const Vec3d& pos) { position.body = pos; return *this; }
    void set_body_position(const Vec3d& value) { position.body = value; }
    void set_initial_anchor(const Vec3d& value) { initial.anchor = value; }
    void set_initial_quaternion(const Quaternion& value) { initial.quaternion = value; }
    void set_initial_pib(const Point2d& p) { initial.pib = p; }
    void set_isRef(bool b) { initial.isRef = b; }
    void set_noMatch(int value) { noMatch = value; }

   private:
    frame position;
    int noMatch = 0;
    int ID = 0;
};
typedef std::vector<Feature> active;
typedef active::iterator Fiter;
typedef active::const_iterator cFiter;

// include/view.hpp:11-30
struct vfeat {
    vfeat(const Point2d& cur, const Point2d& init, const Point2d& refl) : current(cur), initial(init), reflection(refl) {}
    vfeat(const Point2d& cur, const Point2d& init) : current(cur), initial(init), reflection(NONREF) {}
    vfeat(const Vec3d& cur, const Vec3d& init, const Vec3d& refl)
        : current(Point2d(cur[0], cur[1])), initial(Point2d(init[0], init[1])), reflection(Point2d(refl[0], refl[1])) {}
    vfeat(const Vec3d& cur, const Vec3d& init)
        : current(Point2d(cur[0], cur[1])), initial(Point2d(init[0], init[1])), reflection(NONREF) {}
    Point2d current, initial, reflection;
};
typedef struct vfeat Vfeat;

// include/view.hpp:38-66, src/ARC/view.cpp:3-47: the measurement vector as the reference orders it (altitude, then per
// feature current, initial and -- only when its x is not 0 -- reflection); `toMat` flattens into a std::vector here.
class View {
   public:
    View() : altitude(0) {}
    void toMat(std::vector<double>& R) const
    {
        R.clear();
        R.push_back(altitude);
        for (std::vector<Vfeat>::const_iterator it = features.begin(); it != features.end(); ++it) {
            R.push_back(it->current.x);
            R.push_back(it->current.y);
            R.push_back(it->initial.x);
            R.push_back(it->initial.y);
            if (it->reflection.x != 0) {           // features with no reflection add no rows
                R.push_back(it->reflection.x);
                R.push_back(it->reflection.y);
            }
        }
    }
    double altitude;
    std::vector<Vfeat> features;
};
inline void subtract(const View& lhs, const View& rhs, View& dst)
{
    dst.altitude = lhs.altitude - rhs.altitude;
    std::vector<Vfeat>::const_iterator lt = lhs.features.begin(), rt = rhs.features.begin();
    for (; lt != lhs.features.end(); ++lt, ++rt) {
        Point2d reflection(0, 0);
        if (lt->reflection.x != 0) reflection = lt->reflection - rt->reflection;
        dst.features.push_back(Vfeat(lt->current - rt->current, lt->initial - rt->initial, reflection));
    }
}

// include/states.h:21-59 -- the state lives on the GPU; this object is the handle plus the host-visible
// members of the reference's class (X, V, b, features, rf, nrf).  Errors print the reference's message and exit(1).
// A States without a handle is the plain value the reference passes around (the increment of `dynamics`, the correction
// `States(kx)`): X, V, b and the body positions of `features`, with the reference's arithmetic on them.
class States {
   public:
    States() {}
    explicit States(int max_features)
    {
        slamk_params p;
        slamk_default_params(&p);
        if (slamk_create(&p, max_features, 1, 0, &h) != SLAMK_OK) die("slamk_create");
    }
    // the correction K e as a state (src/ARC/states.cpp:24-37): rows 0..2 X, 3..5 V, 6 + 3 i .. the body position of
    // feature i (ID 0); the last three rows (the bias block) are ignored, b stays zero
    explicit States(const std::vector<double>& kx) : X(kx[0], kx[1], kx[2]), V(kx[3], kx[4], kx[5])
    {
        const int nf = ((int)kx.size() - 9) / 3;
        for (int i = 0; i < nf; ++i) addFeature(Feature(Vec3d(kx[6 + 3 * i], kx[7 + 3 * i], kx[8 + 3 * i]), 0));
    }
    States(const States& o) : X(o.X), V(o.V), b(o.b), features(o.features), rf(o.rf), nrf(o.nrf) {}   // a value copy owns no handle
    States& operator=(const States& o)
    {
        X = o.X; V = o.V; b = o.b; features = o.features; rf = o.rf; nrf = o.nrf;
        return *this;
    }
    ~States() { if (h) slamk_destroy(h); }
    States& setX(const Vec3d& pos) { X = pos; return *this; }
    States& setV(const Vec3d& vel) { V = vel; return *this; }
    States& setb(const Vec3d& bias) { b = bias; return *this; }
    States& addFeature(const Feature& f) { features.push_back(f); return *this; }
    // states.cpp:69-86
    void add(const States& a)
    {
        if (a.features.size() != features.size()) { fprintf(stderr, "add: feature mismatch\n"); exit(EXIT_FAILURE); }
        X += a.X;
        V += a.V;
        for (size_t i = 0; i < features.size(); ++i)
            features[i].set_body_position(features[i].get_body_position() + a.features[i].get_body_position());
        b += a.b;
    }
    States& operator+=(const States& rhs) { add(rhs); return *this; }               // states.cpp:17-22
    States& operator*=(const double& rhs)                                           // states.cpp:6-14 (b is not scaled)
    {
        X *= rhs;
        V *= rhs;
        for (Fiter fi = features.begin(); fi != features.end(); ++fi) fi->set_body_position(fi->get_body_position() * rhs);
        return *this;
    }
    // f(mu) dt of the motion model on the host mirror (states.cpp:222-263; dynamics_kernel is the device form): the
    // increment that `mu += mu.dynamics(sense, dt)` adds.  Gravity and the -w x V term are commented out in the reference.
    States dynamics(const Sensors& s, double dt) const
    {
        States f;
        const Matx33d Rb2w = s.quat.get_value().rotation();
        const Vec3d w = s.ang.get_value();
        f.X = Rb2w * V;
        f.V = s.acc.get_value();
        for (cFiter it = features.begin(); it != features.end(); ++it) {
            const Vec3d bp = it->get_body_position();
            Feature fi;
            fi.set_body_position(Vec3d(
                (-V[1] + bp[0] * V[0]) * bp[2] + bp[1] * w[0] - (1 + bp[0] * bp[0]) * w[2] + bp[0] * bp[1] * w[1],
                (-V[2] + bp[1] * V[0]) * bp[2] - bp[0] * w[0] + (1 + bp[1] * bp[1]) * w[1] - bp[0] * bp[1] * w[2],
                (-w[2] * bp[0] + w[1] * bp[1]) * bp[2] + V[0] * bp[2] * bp[2]));
            f.addFeature(fi);
        }
        f.V -= b;
        f.b = Vec3d(0, 0, 0);
        f *= dt;
        return f;
    }
    Vec3d X, V, b;
    active features;                 // mirror of the device containers, refreshed by pull_features()
    int rf = 0, nrf = 0;             // features initialised with / without a reflection (states.cpp:203-211)
    Vec3d getX() { return X; }
    Vec3d getV() { return V; }
    Vec3d getb() { return b; }
    // Downloads the feature containers: slot order, ids and values as the reference's `features` vector holds them.
    void pull_features()
    {
        const int nf = getNumFeatures();
        std::vector<int32_t> id(nf + 1);
        std::vector<double> pib(3 * nf + 1), anchor(3 * nf + 1), q0(4 * nf + 1), pib0(2 * nf + 1);
        std::vector<uint8_t> isr(nf + 1);
        int n = 0;
        if (slamk_get_features(h, 0, &n, id.data(), pib.data(), anchor.data(), q0.data(), pib0.data(), isr.data()))
            die("get_features");
        features.assign(nf, Feature());
        rf = nrf = 0;
        for (int i = 0; i < nf; ++i) {
            Feature& f = features[i];
            f.setID(id[i]);
            f.set_body_position(Vec3d(pib[3 * i], pib[3 * i + 1], pib[3 * i + 2]));
            f.initial.anchor = Vec3d(anchor[3 * i], anchor[3 * i + 1], anchor[3 * i + 2]);
            f.initial.quaternion = Quaternion(q0[4 * i], q0[4 * i + 1], q0[4 * i + 2], q0[4 * i + 3]);
            f.initial.pib = Point2d(pib0[2 * i], pib0[2 * i + 1]);
            f.initial.inverse_depth = pib[3 * i + 2];
            f.initial.isRef = isr[i] != 0;
            if (f.initial.isRef) ++rf; else ++nrf;
        }
    }
    int getNumFeatures() const { return h ? slamk_num_features(h, 0) : (int)features.size(); }
    int getRows() const { return 9 + 3 * getNumFeatures(); }
    void pushX() { if (slamk_set_state(h, 0, X.v, nullptr, nullptr)) die("set_state"); }
    void pushV() { if (slamk_set_state(h, 0, nullptr, V.v, nullptr)) die("set_state"); }
    void pull() { if (slamk_get_state(h, 0, X.v, V.v, b.v)) die("get_state"); }
    // States::update_features(imgsense, sense, P): association + covariance surgery on the device
    void update_features(const ImageSensor& imgsense, const Sensors& sense)
    {
        std::vector<slamk_match> m(imgsense.matches.size());
        for (size_t i = 0; i < m.size(); ++i) {
            const projection& p = imgsense.matches[i];
            m[i].id = p.id; m[i].sx = p.source.x; m[i].sy = p.source.y; m[i].rx = p.reflection.x; m[i].ry = p.reflection.y;
        }
        const Quaternion q = sense.quat.get_value();
        int rc = slamk_frame(h, 0, m.data(), (int)m.size(), q.coord.v);
        if (rc == SLAMK_ENOMATCH) { fprintf(stderr, "Cannot find match.\n"); exit(EXIT_FAILURE); }
        if (rc) die("slamk_frame");
    }
    // F, mu += dynamics*dt, P = F P F^T + Q     (main.cpp:192-194,204-206)
    void predict(const Sensors& sense, double dt)
    {
        const Quaternion q = sense.quat.get_value();
        const Vec3d w = sense.ang.get_value(), a = sense.acc.get_value();
        if (slamk_predict(h, q.coord.v, w.v, a.v, &dt)) die("slamk_predict");
    }
    // measurement model, gain, covariance and gated mean update     (main.cpp:198-202,209-233)
    void update(const Sensors& sense)
    {
        const Quaternion q = sense.quat.get_value();
        const double alt = sense.alt.get_value();
        int applied = 0;
        int rc = slamk_update(h, q.coord.v, &alt, &applied);
        if (rc) die("slamk_update");
    }
    slamk_handle* h = nullptr;

   private:
    [[noreturn]] void die(const char* what)
    {
        fprintf(stderr, "%s: %s\n", what, h ? slamk_last_error(h) : "no handle");
        exit(EXIT_FAILURE);
    }
};

// include/states.h:60-70
inline States operator*(States lhs, const double& rhs) { lhs *= rhs; return lhs; }
inline States operator+(States lhs, const States& rhs) { lhs += rhs; return lhs; }

}  // namespace slamhost
