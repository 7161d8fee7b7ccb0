// elm_ekf.cpp -- plain-CPU counterpart of the reference's 27-state EKF pose update + IMU prediction (SURVEY.md 8 row
// f1).  north_star keeps this filter on the CPU; it is here so that the full stream of BASELINE config 5 (deskew + ICP on
// the GPU, EKF update on the CPU) can be closed: the ICP pose feeds RunGnssUpdate, the EKF pose seeds the next ICP.
// Follows ekf_localization/src/ekf_algorithm.cpp (Init :23-69, RunPredictionImu :167-316, RunGnssUpdate :318-432,
// ComplementaryKalmanFilter :597-700, GetCurrentState :778-833), ekf_algorithm.hpp (UpdateEkfState :116-145, Check*
// :148-213), ekf_localization.cpp (CallbackPcmOdom :147-179, GnssTimeCompensation :323-394, PublishInThread :397-410) and
// localization_functions.hpp (RotToVec :312-333, CalEulerResidualFromQuat :354-370, Exp :412-419, ExpGyroToQuat :439-443,
// PartialDerivativeRotWrtGyro :466-483, ConvertGlobalToLocalVelocity :491-513).  The side modes that the shipped localization.ini
// leaves off are here too: RunPrediction (the constant-velocity model of use_imu = 0, ekfa.cpp:81-165), RunCanUpdate + ZuptCan
// (:434-506, :567-587), ZuptImu (:508-565) and CalibrateVehicleToImu (:703-776); and the node's NavSatFix front end (ekfl.cpp:92-125,
// 643-648) with GeographicLib's LocalCartesian::Forward restated from its published formulas (WGS84 geodetic -> ECEF -> east-north-up).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <deque>

#include "../../include/elimaloc_hip.h"

namespace {
constexpr int N = 27; // STATE_ORDER
enum { S_X = 0, S_Y, S_Z, S_ROLL, S_PITCH, S_YAW, S_VX, S_VY, S_VZ, S_ROLL_RATE, S_PITCH_RATE, S_YAW_RATE, S_AX, S_AY, S_AZ,
       S_B_ROLL_RATE, S_B_PITCH_RATE, S_B_YAW_RATE, S_B_AX, S_B_AY, S_B_AZ, S_G_X, S_G_Y, S_G_Z, S_IMU_ROLL, S_IMU_PITCH, S_IMU_YAW };
constexpr double INIT_STATE_COV = 100.0;

struct V3 { double x, y, z; };
struct Q { double w, x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double norm(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

Q q_mul(Q a, Q b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
Q q_normalized(Q q) {
    const double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    return {q.w / n, q.x / n, q.y / n, q.z / n};
}
Q q_from_angle_axis(double angle, V3 axis) { // Quaterniond(AngleAxisd(angle, axis))
    const double h = 0.5 * angle, s = sin(h);
    return {cos(h), s * axis.x, s * axis.y, s * axis.z};
}
Q q_from_rotvec(V3 v) { // Quaterniond(AngleAxisd(v.norm(), v.normalized()))
    const double n = norm(v);
    V3 ax = v;
    if (n * n > 0.0) ax = v * (1.0 / n);
    return q_from_angle_axis(n, ax);
}
void q_to_R(Q q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
Q q_from_R(const double m[9]) {
    Q q;
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
        v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
V3 q_inv_rotate(Q q, V3 v) { // S_.rot.inverse() * v
    const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    const Q c{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
    const V3 qv{c.x, c.y, c.z};
    V3 uv = cross(qv, v);
    uv = uv + uv;
    return v + uv * c.w + cross(qv, uv);
}
V3 q_rotate(Q q, V3 v) { // Quaterniond * Vector3d (Eigen's _transformVector: v + 2 w (u x v) + 2 u x (u x v))
    const V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v);
    uv = uv + uv;
    return v + uv * q.w + cross(qv, uv);
}
Q q_inverse(Q q) { // Quaterniond::inverse(): conjugate / squaredNorm
    const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
V3 R_mul(const double R[9], V3 v) { return {R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z}; }

void so3_exp(V3 omega, double R[9]) { // lf.hpp:412-419
    const double theta = norm(omega);
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (theta < 1e-5) return;
    const V3 a = omega * (1.0 / theta);
    const double K[9] = {0, -a.z, a.y, a.z, 0, -a.x, -a.y, a.x, 0};
    double KK[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) KK[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
    const double s = sin(theta), c1 = 1.0 - cos(theta);
    for (int i = 0; i < 9; ++i) R[i] += s * K[i] + c1 * KK[i];
}
Q exp_gyro_to_quat(V3 gyro, double dt) { // lf.hpp:439-443
    double R[9];
    so3_exp(gyro * dt, R);
    return q_from_R(R);
}
void partial_rot_wrt_gyro(V3 gyro, double dt, double J[9]) { // lf.hpp:466-483
    const V3 omega = gyro * dt;
    const double theta = norm(omega);
    for (int i = 0; i < 9; ++i) J[i] = 0.0;
    if (theta < 1e-5) return;
    const V3 a = omega * (1.0 / theta);
    const double K[9] = {0, -a.z, a.y, a.z, 0, -a.x, -a.y, a.x, 0};
    double KK[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) KK[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
    const double c1 = (1 - cos(theta)) / (theta * theta), c2 = (theta - sin(theta)) / (theta * theta * theta);
    for (int i = 0; i < 9; ++i) J[i] = dt * (((i % 4 == 0) ? 1.0 : 0.0) + c1 * K[i] + c2 * KK[i]);
}
V3 rot_to_vec(const double R[9]) { // lf.hpp:312-333
    V3 a;
    if (fabs(R[6]) > 0.998) {
        a.z = atan2(-R[5], R[4]);
        a.y = M_PI / 2 * (R[6] >= 0 ? 1 : -1);
        a.x = 0;
    } else {
        a.y = asin(-R[6]);
        a.x = atan2(R[7] / cos(a.y), R[8] / cos(a.y));
        a.z = atan2(R[3] / cos(a.y), R[0] / cos(a.y));
    }
    a.x = fmod(a.x + M_PI, 2 * M_PI) - M_PI;
    a.y = fmod(a.y + M_PI, 2 * M_PI) - M_PI;
    a.z = fmod(a.z + M_PI, 2 * M_PI) - M_PI;
    return a;
}
double norm_angle(double a) { // lf.hpp:263-271
    while (a > M_PI) a -= M_PI * 2.;
    while (a < -M_PI) a += M_PI * 2.;
    return a;
}
double angle_diff(double ref, double rel) { // lf.hpp:295-303
    double d = rel - ref;
    while (d > M_PI) d -= 2. * M_PI;
    while (d < -M_PI) d += 2. * M_PI;
    return d;
}
void global_to_local(double gx, double gy, double gz, double roll, double pitch, double yaw, double* lx, double* ly, double* lz) {
    const double cy = cos(yaw), sy = sin(yaw), cp = cos(pitch), sp = sin(pitch), cr = cos(roll), sr = sin(roll);
    *lx = gx * (cy * cp) + gy * (sy * cp) + gz * (-sp);
    *ly = gx * (cy * sp * sr - sy * cr) + gy * (sy * sp * sr + cy * cr) + gz * (cp * sr);
    *lz = gx * (cy * sp * cr + sy * sr) + gy * (sy * sp * cr - cy * sr) + gz * (cp * cr);
}
// general inverse of an m x m matrix (m <= 6), Gauss-Jordan with partial pivoting
void inv_small(const double* A, int m, double* R) {
    double a[36];
    memcpy(a, A, sizeof(double) * m * m);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) R[i * m + j] = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < m; ++k) {
        int p = k;
        for (int i = k + 1; i < m; ++i)
            if (fabs(a[i * m + k]) > fabs(a[p * m + k])) p = i;
        if (p != k)
            for (int j = 0; j < m; ++j) { std::swap(a[k * m + j], a[p * m + j]); std::swap(R[k * m + j], R[p * m + j]); }
        const double piv = 1.0 / a[k * m + k];
        for (int j = 0; j < m; ++j) { a[k * m + j] *= piv; R[k * m + j] *= piv; }
        for (int i = 0; i < m; ++i) {
            if (i == k) continue;
            const double f = a[i * m + k];
            for (int j = 0; j < m; ++j) { a[i * m + j] -= f * a[k * m + j]; R[i * m + j] -= f * R[k * m + j]; }
        }
    }
}
} // namespace

struct elm_ekf {
    elm_ekf_config cfg;
    V3 pos, vel, gyro, acc, bg, ba, grav;
    Q rot, imu_rot;
    double P[N * N];
    bool reset_for_init_prediction = true, state_initialized = false, yaw_initialized = false, rotation_stabilized = false,
         state_stabilized = false, pcm_init_on_going = false;
    int pcm_update_count = 0;
    double prev_timestamp = 0.0, prev_gnss_timestamp = 0.0;
    // function-static variables of ComplementaryKalmanFilter (ekfa.cpp:613-614)
    bool ckf_started = false;
    double ckf_prev_vel_local_x = 0.0, ckf_prev_time = 0.0;
    elm_ego_state prev_ego{};
    std::deque<elm_ego_state> deq_ekf_state; // ekfl.hpp:136
    double prev_can_timestamp = 0.0;    // prev_can_ (memset to zero in the constructor, ekfa.cpp:17)
    double can_yaw_rate_bias_rad = 0.0; // d_can_yaw_rate_bias_rad_ (ekfa.hpp:279)
    bool vehicle_imu_calib_started = false;
};

static void ekf_init(elm_ekf* e) { // EkfAlgorithm::Init (ekfa.cpp:23-69)
    const elm_ekf_config& c = e->cfg;
    e->pos = {c.ekf_init_x_m, c.ekf_init_y_m, c.ekf_init_z_m};
    e->rot = q_mul(q_mul(q_from_angle_axis(c.ekf_init_yaw_deg * M_PI / 180.0, {0, 0, 1}), q_from_angle_axis(c.ekf_init_pitch_deg * M_PI / 180.0, {0, 1, 0})),
                   q_from_angle_axis(c.ekf_init_roll_deg * M_PI / 180.0, {1, 0, 0}));
    e->imu_rot = {1, 0, 0, 0};
    e->vel = e->gyro = e->acc = e->bg = e->ba = {0, 0, 0};
    e->grav = {0, 0, c.imu_gravity};
    for (int i = 0; i < N * N; ++i) e->P[i] = 0.0;
    for (int i = 0; i < N; ++i) e->P[i * N + i] = INIT_STATE_COV;
    for (int i = S_B_ROLL_RATE; i <= S_B_YAW_RATE; ++i) e->P[i * N + i] = c.ekf_imu_bias_cov_gyro;
    for (int i = S_B_AX; i <= S_B_AZ; ++i) e->P[i * N + i] = c.ekf_imu_bias_cov_acc;
    for (int i = S_G_X; i <= S_G_Z; ++i) e->P[i * N + i] = c.ekf_imu_bias_cov_acc;
    for (int i = S_IMU_ROLL; i <= S_IMU_YAW; ++i) e->P[i * N + i] = c.ekf_imu_bias_cov_gyro;
    e->reset_for_init_prediction = true;
    e->yaw_initialized = e->state_initialized = e->rotation_stabilized = e->state_stabilized = e->pcm_init_on_going = false;
}

// ZuptImu (ekfa.cpp:508-565): near standstill the velocity is pulled to zero and the IMU biases (and gravity) follow the raw readings
static void zupt_imu(elm_ekf* e, V3 imu_gyro, V3 imu_acc) {
    const double alpha = 0.01, gamma = 0.01, vel_thre = 0.1, gyro_thre = 0.1, acc_thre = 0.1;
    const V3 vel_local = q_inv_rotate(e->rot, e->vel);
    if (fabs(vel_local.x) > vel_thre) return;
    const double vel_coeff = (vel_thre - fabs(vel_local.x)) / vel_thre * 0.1; // head<1>().norm()
    const V3 vel_error = e->vel * -1.0;
    e->vel = e->vel + vel_error * vel_coeff;
    const double acc_xy = sqrt(e->acc.x * e->acc.x + e->acc.y * e->acc.y);
    if (norm(e->gyro) > gyro_thre || acc_xy > acc_thre) return;
    const V3 gyro_error = imu_gyro - e->bg;
    e->bg = e->bg + gyro_error * gamma;
    const V3 grav_local = q_inv_rotate(e->rot, e->grav);
    const V3 acc_error_loc = imu_acc - (grav_local + e->ba);
    const V3 acc_error_global = q_rotate(e->rot, imu_acc - e->ba) - e->grav; // with the bias BEFORE its correction
    e->ba = e->ba + acc_error_loc * alpha;
    if (e->cfg.imu_estimate_gravity) e->grav.z += alpha * acc_error_global.z;
}

static void kalman_update(elm_ekf* e, const int* idx, int m, const double* Rm, const double* Y);

// CalibrateVehicleToImu (ekfa.cpp:703-776): above 3 m/s and with a stabilised attitude the direction of travel seen from the IMU
// frame is an observation of the IMU mounting angles (pitch, yaw; roll is not observable and gets a zero innovation)
static void calibrate_vehicle_to_imu(elm_ekf* e) {
    const V3 vel = e->vel;
    if (norm(vel) < 3.0) return;
    if (!e->rotation_stabilized) return;
    e->vehicle_imu_calib_started = true;
    const Q rel = q_mul(e->rot, q_inverse(e->imu_rot));
    const V3 vl = q_rotate(q_inverse(rel), vel);
    const double n = norm(vl);
    V3 dir = vl;
    if (n * n > 0.0) dir = vl * (1.0 / n);
    const double d_yaw = atan2(dir.y, dir.x), d_pitch = -asin(dir.z), d_roll = 0.0;
    const double innovation[3] = {-d_roll, -d_pitch, -d_yaw};
    const double r1 = pow(1.0 * M_PI / 180.0, 2); // the dynamic uncertainty the reference computes is overwritten by these
    const double R3[9] = {r1, 0, 0, 0, r1, 0, 0, 0, r1};
    const int idx[3] = {S_IMU_ROLL, S_IMU_PITCH, S_IMU_YAW};
    kalman_update(e, idx, 3, R3, innovation);
}

static inline double Pd(const elm_ekf* e, int i) { return e->P[i * N + i]; }
static void check_flags(elm_ekf* e, bool yaw, bool init, bool rot, bool stab) { // ekfa.hpp:148-213
    const double d5 = 5.0 * M_PI / 180.0, d02 = 0.2 * M_PI / 180.0;
    if (yaw) e->yaw_initialized = sqrt(Pd(e, S_YAW)) < d5;
    if (init) e->state_initialized = sqrt(Pd(e, S_ROLL)) < d5 && sqrt(Pd(e, S_PITCH)) < d5 && sqrt(Pd(e, S_YAW)) < d5 && sqrt(Pd(e, S_X)) < 1.0 && sqrt(Pd(e, S_Y)) < 1.0;
    if (rot) e->rotation_stabilized = sqrt(Pd(e, S_ROLL)) < d02 && sqrt(Pd(e, S_PITCH)) < d02 && sqrt(Pd(e, S_YAW)) < d02;
    if (stab) e->state_stabilized = sqrt(Pd(e, S_ROLL)) < d02 && sqrt(Pd(e, S_PITCH)) < d02 && sqrt(Pd(e, S_YAW)) < d02 && sqrt(Pd(e, S_X)) < 0.5 && sqrt(Pd(e, S_Y)) < 0.5;
}

// UpdateEkfState (ekfa.hpp:116-145) with H selecting the state rows idx[0..m): K = P H^T S^-1, X += K Y, P -= K H P
static void kalman_update(elm_ekf* e, const int* idx, int m, const double* Rm, const double* Y) {
    double S[36], Sinv[36], PHt[N * 6], K[N * 6];
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) S[i * m + j] = e->P[idx[i] * N + idx[j]] + Rm[i * m + j];
    inv_small(S, m, Sinv);
    for (int r = 0; r < N; ++r)
        for (int j = 0; j < m; ++j) PHt[r * m + j] = e->P[r * N + idx[j]];
    for (int r = 0; r < N; ++r)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int k = 0; k < m; ++k) s += PHt[r * m + k] * Sinv[k * m + j];
            K[r * m + j] = s;
        }
    double su[N];
    for (int r = 0; r < N; ++r) {
        double s = 0.0;
        for (int j = 0; j < m; ++j) s += K[r * m + j] * Y[j];
        su[r] = s;
    }
    e->pos = e->pos + V3{su[0], su[1], su[2]};
    e->vel = e->vel + V3{su[S_VX], su[S_VY], su[S_VZ]};
    e->gyro = e->gyro + V3{su[S_ROLL_RATE], su[S_PITCH_RATE], su[S_YAW_RATE]};
    e->acc = e->acc + V3{su[S_AX], su[S_AY], su[S_AZ]};
    e->bg = e->bg + V3{su[S_B_ROLL_RATE], su[S_B_PITCH_RATE], su[S_B_YAW_RATE]};
    e->ba = e->ba + V3{su[S_B_AX], su[S_B_AY], su[S_B_AZ]};
    e->grav = e->grav + V3{su[S_G_X], su[S_G_Y], su[S_G_Z]};
    e->rot = q_normalized(q_mul(e->rot, q_from_rotvec({su[3], su[4], su[5]})));
    e->imu_rot = q_normalized(q_mul(e->imu_rot, q_from_rotvec({su[24], su[25], su[26]})));
    // P = P - K * (H P)
    double HP[6 * N];
    for (int j = 0; j < m; ++j)
        for (int c = 0; c < N; ++c) HP[j * N + c] = e->P[idx[j] * N + c];
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += K[r * m + j] * HP[j * N + c];
            e->P[r * N + c] -= s;
        }
}

static void complementary_filter(elm_ekf* e, double timestamp, V3 imu_acc) { // ekfa.cpp:597-700
    const V3 acc_meas = imu_acc - e->ba;
    const V3 vel_local = q_inv_rotate(e->rot, e->vel);
    const double centripetal_acc = vel_local.x * e->gyro.z;
    if (!e->ckf_started) { // first call initialises the function-statics with the current values
        e->ckf_started = true;
        e->ckf_prev_vel_local_x = vel_local.x;
        e->ckf_prev_time = timestamp;
    }
    const double dt = timestamp - e->ckf_prev_time;
    if (dt < 1e-6) return;
    const double est_acc_x = (vel_local.x - e->ckf_prev_vel_local_x) / dt;
    e->ckf_prev_vel_local_x = vel_local.x;
    e->ckf_prev_time = timestamp;
    V3 comp = acc_meas - V3{0, centripetal_acc, 0};
    if (e->rotation_stabilized) comp = comp - V3{est_acc_x, 0, 0};
    const double acc_diff = norm(acc_meas) - norm(e->grav);
    const double cn = norm(comp);
    V3 gdir = comp;
    if (cn * cn > 0.0) gdir = comp * (1.0 / cn);
    const double z0 = atan2(gdir.y, gdir.z), z1 = -asin(gdir.x);
    double Rm[9];
    q_to_R(e->rot, Rm);
    const V3 rpy = rot_to_vec(Rm);
    double innovation[2] = {norm_angle(z0 - rpy.x), norm_angle(z1 - rpy.y)};
    double base = 1.0 * M_PI / 180.0;
    if (!e->state_initialized) base = 10.0 * M_PI / 180.0;
    const double cu = fabs(centripetal_acc) / 9.81 * 10.0, lu = fabs(est_acc_x) / 9.81 * 10.0, du = fabs(acc_diff) / 9.81 * 10.0;
    const double lat = 1.0 + du + cu, lon = 1.0 + du + lu;
    const double floor_ = pow(1.0 * M_PI / 180.0, 2);
    const double R2[4] = {std::max(pow(base * lat, 2), floor_), 0, 0, std::max(pow(base * lon, 2), floor_)};
    const int idx[2] = {S_ROLL, S_PITCH};
    kalman_update(e, idx, 2, R2, innovation);
}

extern "C" void elm_ekf_config_default(elm_ekf_config* c) { // config/localization.ini [ekf_localization]
    memset(c, 0, sizeof(*c));
    c->imu_gravity = 9.81; c->imu_estimate_gravity = 1; c->imu_estimate_calibration = 0; c->use_zupt = 0;
    c->use_complementary_filter = 1; c->gps_type = 2;
    c->ekf_init_x_m = -11.2379716113593; c->ekf_init_y_m = 2.61441970033043; c->ekf_init_z_m = -0.576108843465071;
    c->ekf_init_roll_deg = -0.1147311; c->ekf_init_pitch_deg = -0.4350557; c->ekf_init_yaw_deg = -129.8725387;
    c->state_std_pos_m = 0.02; c->state_std_rot_deg = 0.2; c->state_std_vel_mps = 2.0; c->state_std_gyro_dps = 5.0; c->state_std_acc_mps = 100.0;
    c->imu_std_gyro_dps = 0.01; c->imu_std_acc_mps = 0.001; c->ekf_imu_bias_cov_gyro = 0.0001; c->ekf_imu_bias_cov_acc = 0.0001;
    c->gnss_min_cov_x_m = 0.2; c->gnss_min_cov_y_m = 0.2; c->gnss_min_cov_z_m = 0.7;
    c->can_vel_scale_factor = 1.0; c->ekf_can_meas_uncertainty_vel_mps = 2.0; c->ekf_can_meas_uncertainty_yaw_rate_deg = 10.0;
}

extern "C" int elm_ekf_create(const elm_ekf_config* cfg, elm_ekf** out) {
    if (!cfg || !out) return ELM_ERR_INVALID;
    elm_ekf* e = new elm_ekf();
    e->cfg = *cfg;
    ekf_init(e);
    *out = e;
    return ELM_OK;
}
extern "C" void elm_ekf_destroy(elm_ekf* e) { delete e; }

extern "C" int elm_ekf_predict_imu(elm_ekf* e, double t, const double gyro_in[3], const double acc_in[3], int* predicted) {
    if (!e || !gyro_in || !acc_in || !predicted) return ELM_ERR_INVALID;
    *predicted = 0;
    const elm_ekf_config& c = e->cfg;
    const V3 ig{gyro_in[0], gyro_in[1], gyro_in[2]}, ia{acc_in[0], acc_in[1], acc_in[2]};
    if (e->reset_for_init_prediction) { e->prev_timestamp = t; e->reset_for_init_prediction = false; return ELM_OK; }
    if (e->pcm_init_on_going) { e->prev_timestamp = t; return ELM_OK; }
    check_flags(e, false, false, true, false);
    const bool use_ckf = (c.gps_type == 1 /*BESTPOS*/ || c.use_complementary_filter);
    if (!e->state_initialized) {
        e->prev_timestamp = t;
        if (e->yaw_initialized && use_ckf) complementary_filter(e, t, ia);
        return ELM_OK;
    }
    if (fabs(t - e->prev_timestamp) < 1e-6) return ELM_OK;
    const double dt = t - e->prev_timestamp;
    const V3 pbg = e->bg, pba = e->ba, pgrav = e->grav, pvel = e->vel;
    const Q prot = e->rot;
    double G[9];
    q_to_R(e->rot, G);
    const V3 cg = ig - pbg;
    e->rot = q_normalized(q_mul(prot, exp_gyro_to_quat(cg, dt)));
    const V3 ca = ia - pba;
    const V3 ag = R_mul(G, ca) - pgrav;
    e->pos = e->pos + pvel * dt + ag * (0.5 * dt * dt);
    e->vel = e->vel + ag * dt;
    e->gyro = cg;
    e->acc = ag;
    // F and Q (ekfa.cpp:255-300)
    static thread_local double F[N * N], FP[N * N];
    for (int i = 0; i < N * N; ++i) F[i] = 0.0;
    for (int i = 0; i < N; ++i) F[i * N + i] = 1.0;
    double J[9];
    partial_rot_wrt_gyro(cg, dt, J);
    for (int r = 0; r < 3; ++r) {
        F[(S_X + r) * N + (S_VX + r)] = dt;
        for (int cc = 0; cc < 3; ++cc) {
            F[(S_X + r) * N + (S_B_AX + cc)] = -0.5 * G[r * 3 + cc] * dt * dt;
            F[(S_ROLL + r) * N + (S_B_ROLL_RATE + cc)] = -J[r * 3 + cc];
            F[(S_VX + r) * N + (S_B_AX + cc)] = -G[r * 3 + cc] * dt;
            F[(S_AX + r) * N + (S_B_AX + cc)] = -G[r * 3 + cc];
        }
        F[(S_ROLL_RATE + r) * N + (S_B_ROLL_RATE + r)] = -1.0;
    }
    if (c.imu_estimate_gravity) {
        F[S_Z * N + S_G_Z] = -0.5 * dt * dt;
        F[S_VZ * N + S_G_Z] = -dt;
        F[S_AZ * N + S_G_Z] = -1.0;
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int k = 0; k < N; ++k) s += F[i * N + k] * e->P[k * N + j];
            FP[i * N + j] = s;
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int k = 0; k < N; ++k) s += FP[i * N + k] * F[j * N + k];
            e->P[i * N + j] = s;
        }
    const double d2 = dt * dt, rad = M_PI / 180.0;
    const double qd[9] = {pow(c.state_std_pos_m, 2) * d2, pow(c.state_std_rot_deg * rad, 2) * d2, pow(c.state_std_vel_mps, 2) * d2,
                          pow(c.imu_std_gyro_dps * rad, 2) * d2, pow(c.imu_std_acc_mps, 2) * d2, pow(c.ekf_imu_bias_cov_gyro, 2) * d2,
                          pow(c.ekf_imu_bias_cov_acc, 2) * d2, pow(c.ekf_imu_bias_cov_acc, 2) * d2, pow(c.state_std_rot_deg * rad, 2) * d2};
    for (int b = 0; b < 9; ++b)
        for (int r = 0; r < 3; ++r) e->P[(b * 3 + r) * N + (b * 3 + r)] += qd[b];
    e->prev_timestamp = t;
    if (c.use_zupt) zupt_imu(e, ig, ia); // ekfa.cpp:311-313: ZUPT, complementary filter, mount calibration -- in this order
    if (use_ckf) complementary_filter(e, t, ia);
    if (c.imu_estimate_calibration) calibrate_vehicle_to_imu(e);
    *predicted = 1;
    return ELM_OK;
}

// RunPrediction (ekfa.cpp:81-165): the constant-velocity / constant-acceleration model the node runs from its timer when use_imu = 0
extern "C" int elm_ekf_predict(elm_ekf* e, double t, int* predicted) {
    if (!e || !predicted) return ELM_ERR_INVALID;
    *predicted = 0;
    const elm_ekf_config& c = e->cfg;
    if (e->reset_for_init_prediction) { e->prev_timestamp = t; e->reset_for_init_prediction = false; return ELM_OK; }
    if (e->pcm_init_on_going) { e->prev_timestamp = t; return ELM_OK; }
    if (fabs(t - e->prev_timestamp) < 1e-6) return ELM_OK;
    const double dt = t - e->prev_timestamp;
    const V3 pvel = e->vel, pacc = e->acc, pgyro = e->gyro;
    e->pos = e->pos + (pvel * dt + pacc * (0.5 * dt * dt));
    e->rot = q_normalized(q_mul(e->rot, exp_gyro_to_quat(pgyro, dt)));
    e->vel = e->vel + pacc * dt;
    // P = F P F^T + Q, F = I + dt [pos|vel] + dt [rot|rate] + dt^2 / 2 [pos|acc] + dt [vel|acc]
    static thread_local double F[N * N], FP[N * N];
    for (int i = 0; i < N * N; ++i) F[i] = 0.0;
    for (int i = 0; i < N; ++i) F[i * N + i] = 1.0;
    for (int r = 0; r < 3; ++r) {
        F[(S_X + r) * N + (S_VX + r)] = dt;
        F[(S_ROLL + r) * N + (S_ROLL_RATE + r)] = dt;
        F[(S_X + r) * N + (S_AX + r)] = 0.5 * dt * dt;
        F[(S_VX + r) * N + (S_AX + r)] = dt;
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int k = 0; k < N; ++k) s += F[i * N + k] * e->P[k * N + j];
            FP[i * N + j] = s;
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int k = 0; k < N; ++k) s += FP[i * N + k] * F[j * N + k];
            e->P[i * N + j] = s;
        }
    const double d2 = dt * dt, rad = M_PI / 180.0;
    // (the gyro term is the square of the deg/s figure as it stands: ekfa.cpp:137-138 does not convert it)
    const double qd[5] = {pow(c.state_std_pos_m, 2) * d2, pow(c.state_std_rot_deg * rad, 2) * d2, pow(c.state_std_vel_mps, 2) * d2,
                          pow(c.state_std_gyro_dps, 2) * d2, pow(c.state_std_acc_mps, 2) * d2};
    for (int b = 0; b < 5; ++b)
        for (int r = 0; r < 3; ++r) e->P[(b * 3 + r) * N + (b * 3 + r)] += qd[b];
    e->prev_timestamp = t;
    *predicted = 1;
    return ELM_OK;
}

// RunCanUpdate + ZuptCan (ekfa.cpp:434-506, 567-587): wheel speed (vehicle frame) and yaw rate as a 4-row measurement of the global
// velocity and the yaw rate; the node fills vel.x and gyro.z only (ekfl.cpp:127-137)
extern "C" int elm_ekf_update_can(elm_ekf* e, double t, const double vel_in[3], const double gyro_in[3], int* updated) {
    if (!e || !vel_in || !gyro_in || !updated) return ELM_ERR_INVALID;
    *updated = 0;
    const elm_ekf_config& c = e->cfg;
    const double can_dt = t - e->prev_can_timestamp;
    if (fabs(can_dt) < 0.01) return ELM_OK;
    V3 uvel{vel_in[0], vel_in[1], vel_in[2]};
    const double ugyro_z = gyro_in[2] - e->can_yaw_rate_bias_rad;
    uvel.x *= c.can_vel_scale_factor;
    const V3 vg = q_rotate(e->rot, uvel);
    const double Y[4] = {vg.x - e->vel.x, vg.y - e->vel.y, vg.z - e->vel.z, ugyro_z - e->gyro.z};
    double Rl[3] = {pow(c.ekf_can_meas_uncertainty_vel_mps, 2), pow(c.ekf_can_meas_uncertainty_vel_mps * 2, 2), pow(c.ekf_can_meas_uncertainty_vel_mps * 2, 2)};
    double G[9], R4[16];
    q_to_R(e->rot, G);
    for (int i = 0; i < 16; ++i) R4[i] = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += G[i * 3 + k] * Rl[k] * G[j * 3 + k]; // rot R_local rot^T
            R4[i * 4 + j] = s;
        }
    R4[15] = pow(c.ekf_can_meas_uncertainty_yaw_rate_deg * M_PI / 180.0, 2);
    const int idx[4] = {S_VX, S_VY, S_VZ, S_YAW_RATE};
    kalman_update(e, idx, 4, R4, Y);
    e->prev_can_timestamp = t;
    // ZuptCan with the RAW input
    const double vn = sqrt(vel_in[0] * vel_in[0] + vel_in[1] * vel_in[1] + vel_in[2] * vel_in[2]);
    if (!(vn > 0.05)) {
        const double a = 0.05;
        e->can_yaw_rate_bias_rad = a * gyro_in[2] + (1.0 - a) * e->can_yaw_rate_bias_rad;
        e->vel = e->vel * (1.0 - a);
    }
    *updated = 1;
    return ELM_OK;
}

extern "C" int elm_ekf_update_pose(elm_ekf* e, double t, const double pos[3], const double quat_xyzw[4], const double pos_cov[9],
                                   const double rot_cov[9], int source, int* updated) {
    if (!e || !pos || !quat_xyzw || !pos_cov || !rot_cov || !updated) return ELM_ERR_INVALID;
    *updated = 0;
    const elm_ekf_config& c = e->cfg;
    const Q mq{quat_xyzw[3], quat_xyzw[0], quat_xyzw[1], quat_xyzw[2]};
    if (source == ELM_GNSS_PCM_INIT) { // ekfa.cpp:324-349
        e->pos = {pos[0], pos[1], pos[2]};
        e->rot = mq;
        e->vel = e->gyro = e->acc = e->bg = e->ba = {0, 0, 0};
        e->grav = {0, 0, c.imu_gravity};
        for (int i = 0; i <= S_AZ; ++i)
            for (int j = 0; j <= S_AZ; ++j) e->P[i * N + j] = (i == j) ? INIT_STATE_COV : 0.0;
        e->state_initialized = e->yaw_initialized = true;
        e->pcm_init_on_going = true;
        *updated = 1;
        return ELM_OK;
    }
    check_flags(e, true, true, true, true);
    if (e->pcm_init_on_going && source == ELM_GNSS_PCM) {
        if (e->pcm_update_count > 10) e->pcm_init_on_going = false;
        e->pcm_update_count++;
    }
    double R6[36];
    for (int i = 0; i < 36; ++i) R6[i] = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) { R6[r * 6 + cc] = pos_cov[r * 3 + cc]; R6[(r + 3) * 6 + (cc + 3)] = rot_cov[r * 3 + cc]; }
    const bool gnss_like = (source == ELM_GNSS_NOVATEL || source == ELM_GNSS_BESTPOS || source == ELM_GNSS_NAVSATFIX);
    if (gnss_like) {
        R6[0] += c.gnss_min_cov_x_m; R6[7] += c.gnss_min_cov_y_m; R6[14] += c.gnss_min_cov_z_m;
        R6[21] += c.gnss_min_cov_roll_deg * M_PI / 180.0; R6[28] += c.gnss_min_cov_pitch_deg * M_PI / 180.0; R6[35] += c.gnss_min_cov_yaw_deg * M_PI / 180.0;
    }
    // residual: position difference + Euler-angle difference (CalEulerResidualFromQuat, lf.hpp:354-370)
    double Rs[9], Rm[9];
    q_to_R(q_normalized(e->rot), Rs);
    q_to_R(q_normalized(mq), Rm);
    const V3 sa = rot_to_vec(Rs), ma = rot_to_vec(Rm);
    double Y[6] = {pos[0] - e->pos.x, pos[1] - e->pos.y, pos[2] - e->pos.z, norm_angle(ma.x - sa.x), norm_angle(ma.y - sa.y), norm_angle(ma.z - sa.z)};
    if (source == ELM_GNSS_NAVSATFIX || source == ELM_GNSS_BESTPOS) { // position-only branch (ekfa.cpp:410-424)
        if (!e->yaw_initialized) { R6[0] += 3.0; R6[7] += 3.0; }
        const int idx[3] = {0, 1, 2};
        const double R3[9] = {R6[0], R6[1], R6[2], R6[6], R6[7], R6[8], R6[12], R6[13], R6[14]};
        kalman_update(e, idx, 3, R3, Y);
    } else {
        const int idx[6] = {0, 1, 2, 3, 4, 5};
        kalman_update(e, idx, 6, R6, Y);
    }
    e->prev_gnss_timestamp = t;
    *updated = 1;
    return ELM_OK;
}

// GeographicLib::LocalCartesian(lat0, lon0, h0).Forward(lat, lon, h) (GeographicLib 1.5x, LocalCartesian.cpp / Geocentric.cpp; the
// library is an external dependency of the reference -- ekfl.cpp:643-648 -- and absent here): geodetic -> earth-centred (WGS84:
// a = 6378137 m, f = 1 / 298.257223563), then the rotation to the east / north / up axes at the origin.
static void geocentric(double lat_deg, double lon_deg, double h, double X[3], double M[9]) {
    const double a = 6378137.0, f = 1.0 / 298.257223563, e2 = f * (2.0 - f), e2m = (1.0 - f) * (1.0 - f);
    const double phi = lat_deg * M_PI / 180.0, lam = lon_deg * M_PI / 180.0;
    const double sphi = sin(phi), cphi = cos(phi), slam = sin(lam), clam = cos(lam);
    const double n = a / sqrt(1.0 - e2 * sphi * sphi);
    const double Z = (e2m * n + h) * sphi, R = (n + h) * cphi;
    X[0] = R * clam; X[1] = R * slam; X[2] = Z;
    if (M) { // rows of M^T: east, north, up expressed in the earth-centred frame
        M[0] = -slam;        M[1] = clam;         M[2] = 0.0;
        M[3] = -clam * sphi; M[4] = -slam * sphi; M[5] = cphi;
        M[6] = clam * cphi;  M[7] = slam * cphi;  M[8] = sphi;
    }
}
extern "C" int elm_gps_project(double ref_lat_deg, double ref_lon_deg, double ref_alt_m, double lat_deg, double lon_deg, double alt_m, double xyz[3]) {
    if (!xyz) return ELM_ERR_INVALID;
    double X0[3], X[3], M[9];
    geocentric(ref_lat_deg, ref_lon_deg, ref_alt_m, X0, M);
    geocentric(lat_deg, lon_deg, alt_m, X, nullptr);
    const double d[3] = {X[0] - X0[0], X[1] - X0[1], X[2] - X0[2]};
    for (int i = 0; i < 3; ++i) xyz[i] = (M[i * 3] * d[0] + M[i * 3 + 1] * d[1]) + M[i * 3 + 2] * d[2];
    return ELM_OK;
}
// EkfLocalization::CallbackNavsatFix (ekfl.cpp:92-125): projection, the squared standard deviations of the message's covariance
// diagonal (the node squares them: :104-106), the use_gps switch and the uncertainty gate, then RunGnssUpdate with the NAVSATFIX source
// (position rows only).  pos_out (optional) = the projected position (what the node's GPS marker / odometry shows).
extern "C" int elm_ekf_update_navsatfix(elm_ekf* e, double stamp, double lat_deg, double lon_deg, double alt_m, const double position_covariance[9],
                                        double ref_lat_deg, double ref_lon_deg, double ref_alt_m, int use_gps, double gnss_uncertainty_max_m,
                                        double pos_out[3], int* updated) {
    if (!e || !position_covariance || !updated) return ELM_ERR_INVALID;
    *updated = 0;
    double pos[3];
    elm_gps_project(ref_lat_deg, ref_lon_deg, ref_alt_m, lat_deg, lon_deg, alt_m, pos);
    if (pos_out) memcpy(pos_out, pos, sizeof(pos));
    double pc[9] = {0}, rc[9] = {0};
    pc[0] = pow(position_covariance[0], 2); pc[4] = pow(position_covariance[4], 2); pc[8] = pow(position_covariance[8], 2);
    if (!use_gps) return ELM_OK;
    if (pc[0] > gnss_uncertainty_max_m || pc[4] > gnss_uncertainty_max_m) return ELM_OK;
    const double qi[4] = {0.0, 0.0, 0.0, 1.0}; // NavSatFix carries no attitude: identity, zero covariance
    return elm_ekf_update_pose(e, stamp, pos, qi, pc, rc, ELM_GNSS_NAVSATFIX, updated);
}

extern "C" int elm_ekf_get_state(elm_ekf* e, elm_ekf_state* out) {
    if (!e || !out) return ELM_ERR_INVALID;
    const double v[27] = {e->pos.x, e->pos.y, e->pos.z, 0, 0, 0, e->vel.x, e->vel.y, e->vel.z, e->gyro.x, e->gyro.y, e->gyro.z,
                          e->acc.x, e->acc.y, e->acc.z, e->bg.x, e->bg.y, e->bg.z, e->ba.x, e->ba.y, e->ba.z, e->grav.x, e->grav.y, e->grav.z, 0, 0, 0};
    memcpy(out->x, v, sizeof(v));
    out->rot_xyzw[0] = e->rot.x; out->rot_xyzw[1] = e->rot.y; out->rot_xyzw[2] = e->rot.z; out->rot_xyzw[3] = e->rot.w;
    out->imu_rot_xyzw[0] = e->imu_rot.x; out->imu_rot_xyzw[1] = e->imu_rot.y; out->imu_rot_xyzw[2] = e->imu_rot.z; out->imu_rot_xyzw[3] = e->imu_rot.w;
    memcpy(out->P, e->P, sizeof(e->P));
    out->b_state_initialized = e->state_initialized; out->b_yaw_initialized = e->yaw_initialized;
    out->b_rotation_stabilized = e->rotation_stabilized; out->b_state_stabilized = e->state_stabilized;
    out->b_pcm_init_on_going = e->pcm_init_on_going;
    out->timestamp = e->prev_timestamp;
    return ELM_OK;
}

// GetCurrentState (ekfa.cpp:778-833) + the deque upkeep of PublishInThread (ekfl.cpp:397-410)
extern "C" int elm_ekf_publish(elm_ekf* e, elm_ego_state* out) {
    if (!e || !out) return ELM_ERR_INVALID;
    elm_ego_state s{};
    s.timestamp = e->prev_timestamp;
    if (s.timestamp - e->prev_ego.timestamp < 1e-6) {
        s = e->prev_ego;
    } else {
        s.x_m = e->pos.x; s.y_m = e->pos.y; s.z_m = e->pos.z;
        double R[9];
        q_to_R(e->rot, R);
        const V3 eu = rot_to_vec(R);
        s.roll_rad = eu.x; s.pitch_rad = eu.y; s.yaw_rad = eu.z;
        s.roll_vel = e->gyro.x; s.pitch_vel = e->gyro.y; s.yaw_vel = e->gyro.z;
        global_to_local(e->vel.x, e->vel.y, e->vel.z, eu.x, eu.y, eu.z, &s.vx, &s.vy, &s.vz);
        global_to_local(e->acc.x, e->acc.y, e->acc.z, eu.x, eu.y, eu.z, &s.ax, &s.ay, &s.az);
        global_to_local(Pd(e, S_X), Pd(e, S_Y), Pd(e, S_Z), eu.x, eu.y, eu.z, &s.x_cov_m, &s.y_cov_m, &s.z_cov_m);
        s.x_cov_m = fabs(s.x_cov_m); s.y_cov_m = fabs(s.y_cov_m); s.z_cov_m = fabs(s.z_cov_m);
        s.roll_cov_rad = Pd(e, S_ROLL); s.pitch_cov_rad = Pd(e, S_PITCH); s.yaw_cov_rad = Pd(e, S_YAW);
        e->prev_ego = s;
    }
    if (e->deq_ekf_state.empty() || e->deq_ekf_state.back().timestamp + 1e-5 < s.timestamp) e->deq_ekf_state.push_back(s);
    if (e->deq_ekf_state.back().timestamp > s.timestamp) e->deq_ekf_state.clear();
    while (e->deq_ekf_state.size() > 1000) e->deq_ekf_state.pop_front();
    *out = s;
    return ELM_OK;
}

// CallbackPcmOdom (ekfl.cpp:147-179) -> GnssTimeCompensation (ekfl.cpp:323-394) -> RunGnssUpdate
extern "C" int elm_ekf_update_pcm_odom(elm_ekf* e, double stamp, const double pos[3], const double quat_xyzw[4],
                                       const double covariance36[36], int source, int* updated) {
    if (!e || !pos || !quat_xyzw || !covariance36 || !updated) return ELM_ERR_INVALID;
    *updated = 0;
    double pc[9], rc[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { pc[r * 3 + c] = covariance36[r * 6 + c]; rc[r * 3 + c] = covariance36[(r + 3) * 6 + (c + 3)]; }
    if (source == ELM_GNSS_PCM_INIT) return elm_ekf_update_pose(e, stamp, pos, quat_xyzw, pc, rc, source, updated);
    if (e->deq_ekf_state.empty()) return ELM_OK;
    const elm_ego_state cur = e->deq_ekf_state.back();
    if (e->deq_ekf_state.front().timestamp > stamp) return ELM_OK;
    elm_ego_state closest = e->deq_ekf_state.front();
    for (const auto& s : e->deq_ekf_state) {
        closest = s;
        if (s.timestamp > stamp) break;
    }
    double p[3] = {pos[0], pos[1], pos[2]};
    Q q{quat_xyzw[3], quat_xyzw[0], quat_xyzw[1], quat_xyzw[2]};
    double t_out = stamp;
    const double d_gnss_to_ekf = cur.timestamp - stamp;
    if (d_gnss_to_ekf > 0.0) {
        double dx = 0, dy = 0, dz = 0, d_roll = 0, d_pitch = 0, d_yaw = 0;
        if (fabs(cur.timestamp - closest.timestamp) > 1e-5) {
            const double ratio = d_gnss_to_ekf / (cur.timestamp - closest.timestamp);
            dx = (cur.x_m - closest.x_m) * ratio; dy = (cur.y_m - closest.y_m) * ratio; dz = (cur.z_m - closest.z_m) * ratio;
            d_roll = angle_diff(closest.roll_rad, cur.roll_rad) * ratio;
            d_pitch = angle_diff(closest.pitch_rad, cur.pitch_rad) * ratio;
            d_yaw = angle_diff(closest.yaw_rad, cur.yaw_rad) * ratio;
        }
        t_out = cur.timestamp;
        p[0] += dx; p[1] += dy; p[2] += dz;
        const Q dq = q_mul(q_mul(q_from_angle_axis(d_yaw, {0, 0, 1}), q_from_angle_axis(d_pitch, {0, 1, 0})), q_from_angle_axis(d_roll, {1, 0, 0}));
        q = q_normalized(q_mul(q, dq));
    }
    const double qo[4] = {q.x, q.y, q.z, q.w};
    return elm_ekf_update_pose(e, t_out, p, qo, pc, rc, source, updated);
}
