// elm_io.cpp -- on-disk / wire formats either side of the registration path, host only.
//   * INI reader with the parse rules the reference gets from SimpleIni + IniParser (bsw/system/ini_parser/ini_parser.cpp
//     :41-225): case-insensitive sections/keys, full-line ';' / '#' comments, values keep their trailing "; comment" text and
//     numbers are read with atoi/atof (which stop at the first non-numeric character), bool = atoi(v) > 0, arrays =
//     whitespace-separated stod tokens with "inf"/"-inf"; a later duplicate key replaces the earlier one.
//   * loaders that fill elm_reg_config / elm_pcm_node_config / elm_ekf_config from the reference's own localization.ini +
//     calibration.ini with the key names of pcm.cpp:121-196 and ekfl.cpp:218-316 (missing key = field left unchanged).
//   * PCD map reader (pcl::io::loadPCDFile<PointXYZINormal> as called at pcm.cpp:72-79): ascii / binary / binary_compressed
//     (LZF), x y z float32 picked by field name, every other field ignored.
//   * scan record unpack: PointXYZIT and OusterPointXYZIRT out of a PointCloud2-style byte buffer (pcm.hpp:81-106,
//     pcm.cpp:900-930) including the Ouster index sampling and its trailing default point.
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/elimaloc_hip.h"

namespace {
std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
bool read_file(const char* path, std::string* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    out->clear();
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
    fclose(f);
    return true;
}
} // namespace

struct elm_ini {
    struct Entry { std::string section, key, value; };
    std::vector<Entry> entries;
    const std::string* find(const char* sec, const char* key) const {
        for (const auto& e : entries)
            if (!strcasecmp(e.section.c_str(), sec) && !strcasecmp(e.key.c_str(), key)) return &e.value;
        return nullptr;
    }
};

extern "C" int elm_ini_load(const char* path, elm_ini** out) {
    if (!path || !out) return ELM_ERR_INVALID;
    std::string text;
    if (!read_file(path, &text)) return ELM_ERR_IO;
    elm_ini* ini = new elm_ini();
    if (text.size() >= 3 && (unsigned char)text[0] == 0xEF && (unsigned char)text[1] == 0xBB && (unsigned char)text[2] == 0xBF) text.erase(0, 3);
    std::string section;
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t eol = text.find_first_of("\r\n", pos);
        if (eol == std::string::npos) eol = text.size();
        const std::string line = trim(text.substr(pos, eol - pos));
        pos = eol + 1;
        if (line.empty() || line[0] == ';' || line[0] == '#') continue;
        if (line[0] == '[') {
            const size_t close = line.find(']');
            if (close != std::string::npos) section = trim(line.substr(1, close - 1));
            continue;
        }
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue; // not a key line
        const std::string key = trim(line.substr(0, eq)), value = trim(line.substr(eq + 1));
        bool replaced = false;
        for (auto& e : ini->entries)
            if (!strcasecmp(e.section.c_str(), section.c_str()) && !strcasecmp(e.key.c_str(), key.c_str())) { e.value = value; replaced = true; break; }
        if (!replaced) ini->entries.push_back({section, key, value});
        if (pos > text.size()) break;
    }
    *out = ini;
    return ELM_OK;
}
extern "C" void elm_ini_destroy(elm_ini* ini) { delete ini; }

extern "C" int elm_ini_get_string(const elm_ini* ini, const char* section, const char* key, char* buf, size_t cap) {
    if (!ini || !section || !key || !buf || cap == 0) return ELM_ERR_INVALID;
    const std::string* v = ini->find(section, key);
    if (!v) return 0;
    snprintf(buf, cap, "%s", v->c_str());
    return 1;
}
extern "C" int elm_ini_get_int(const elm_ini* ini, const char* section, const char* key, int* out) {
    if (!ini || !section || !key || !out) return ELM_ERR_INVALID;
    const std::string* v = ini->find(section, key);
    if (!v) return 0;
    *out = atoi(v->c_str());
    return 1;
}
extern "C" int elm_ini_get_bool(const elm_ini* ini, const char* section, const char* key, int* out) {
    int i = 0;
    const int r = elm_ini_get_int(ini, section, key, &i);
    if (r == 1) *out = i > 0 ? 1 : 0;
    return r;
}
extern "C" int elm_ini_get_double(const elm_ini* ini, const char* section, const char* key, double* out) {
    if (!ini || !section || !key || !out) return ELM_ERR_INVALID;
    const std::string* v = ini->find(section, key);
    if (!v) return 0;
    *out = atof(v->c_str());
    return 1;
}
extern "C" int elm_ini_get_array(const elm_ini* ini, const char* section, const char* key, double* out, size_t cap, size_t* n) {
    if (!ini || !section || !key || !n) return ELM_ERR_INVALID;
    *n = 0;
    const std::string* v = ini->find(section, key);
    if (!v) return 0;
    size_t p = 0;
    const std::string& s = *v;
    while (true) {
        while (p < s.size() && isspace((unsigned char)s[p])) ++p;
        if (p >= s.size()) break;
        size_t q = p;
        while (q < s.size() && !isspace((unsigned char)s[q])) ++q;
        const std::string tok = s.substr(p, q - p);
        p = q;
        if (p < s.size() && s[p] == ',') ++p;
        double val;
        if (tok == "inf") val = INFINITY;
        else if (tok == "-inf") val = -INFINITY;
        else {
            char* end = nullptr;
            val = strtod(tok.c_str(), &end);
            if (end == tok.c_str()) return ELM_ERR_INVALID; // std::stod would throw in the reference
        }
        if (out && *n < cap) out[*n] = val;
        ++*n;
    }
    return 1;
}

static void rpy_deg_to_rot(const double rpy_deg[3], double R[9]) { // VecToRot (lf.hpp:340-345): Rz(y) Ry(p) Rx(r), row-major
    const double r = rpy_deg[0] * M_PI / 180.0, p = rpy_deg[1] * M_PI / 180.0, y = rpy_deg[2] * M_PI / 180.0;
    const double cr = cos(r), sr = sin(r), cp = cos(p), sp = sin(p), cy = cos(y), sy = sin(y);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

extern "C" void elm_pcm_node_config_default(elm_pcm_node_config* c) { // loc.ini:2-9, 80-90; cal.ini:15-21
    memset(c, 0, sizeof(*c));
    snprintf(c->lidar_type, sizeof c->lidar_type, "velodyne");
    c->lidar_scan_time_end = 1; c->lidar_time_delay = 0.03; c->pcm_voxel_size = 1.0; c->pcm_voxel_max_point = 30; c->run_deskew = 1;
    c->input_max_dist = 100.0; c->input_index_sampling = 5; c->input_voxel_ds_m = 1.5;
    for (int i = 0; i < 4; ++i) c->tf_ego_to_lidar[i * 4 + i] = 1.0;
}

// ProcessINI (pcm.cpp:121-196).  Either path may be NULL (that file is then skipped).  tf_ego_to_lidar / ego_to_* come
// out in the layouts the rest of the ABI uses: 4x4 column-major, 3x3 column-major.
extern "C" int elm_load_pcm_config(const char* localization_ini, const char* calibration_ini, elm_pcm_node_config* node,
                                   elm_reg_config* reg) {
    if (!node || !reg) return ELM_ERR_INVALID;
    if (calibration_ini) {
        elm_ini* ini = nullptr;
        int st = elm_ini_load(calibration_ini, &ini);
        if (st != ELM_OK) return st;
        double t[3], rl[3], ri[3];
        size_t nt = 0, nl = 0, ni = 0;
        const int a = elm_ini_get_array(ini, "Rear To Main LiDAR", "transform_xyz_m", t, 3, &nt);
        const int b = elm_ini_get_array(ini, "Rear To Main LiDAR", "rotation_rpy_deg", rl, 3, &nl);
        const int c = elm_ini_get_array(ini, "Rear To Imu", "rotation_rpy_deg", ri, 3, &ni);
        elm_ini_destroy(ini);
        if (a < 0 || b < 0 || c < 0 || nt != 3 || nl != 3 || ni != 3) return ELM_ERR_INVALID; // "Invalid Calibration!" (pcm.cpp:144-147)
        double Rl[9], Ri[9];
        rpy_deg_to_rot(rl, Rl);
        rpy_deg_to_rot(ri, Ri);
        for (int r = 0; r < 3; ++r) {
            reg->ego_to_lidar_trans[r] = t[r];
            for (int cc = 0; cc < 3; ++cc) { reg->ego_to_lidar_rot[cc * 3 + r] = Rl[r * 3 + cc]; reg->ego_to_imu_rot[cc * 3 + r] = Ri[r * 3 + cc]; }
        }
        for (int i = 0; i < 16; ++i) node->tf_ego_to_lidar[i] = (i % 5 == 0) ? 1.0 : 0.0;
        for (int r = 0; r < 3; ++r) {
            for (int cc = 0; cc < 3; ++cc) node->tf_ego_to_lidar[cc * 4 + r] = Rl[r * 3 + cc];
            node->tf_ego_to_lidar[12 + r] = t[r];
        }
    }
    if (localization_ini) {
        elm_ini* ini = nullptr;
        int st = elm_ini_load(localization_ini, &ini);
        if (st != ELM_OK) return st;
        elm_ini_get_string(ini, "common_variable", "lidar_type", node->lidar_type, sizeof node->lidar_type);
        elm_ini_get_bool(ini, "common_variable", "lidar_scan_time_end", &node->lidar_scan_time_end);
        elm_ini_get_double(ini, "common_variable", "lidar_time_delay", &node->lidar_time_delay);
        elm_ini_get_bool(ini, "pcm_matching", "debug_print", &reg->b_debug_print);
        elm_ini_get_double(ini, "pcm_matching", "pcm_voxel_size", &node->pcm_voxel_size);
        elm_ini_get_int(ini, "pcm_matching", "pcm_voxel_max_point", &node->pcm_voxel_max_point);
        elm_ini_get_bool(ini, "pcm_matching", "run_deskew", &node->run_deskew);
        elm_ini_get_double(ini, "pcm_matching", "input_max_dist", &node->input_max_dist);
        elm_ini_get_int(ini, "pcm_matching", "input_index_sampling", &node->input_index_sampling);
        elm_ini_get_double(ini, "pcm_matching", "input_voxel_ds_m", &node->input_voxel_ds_m);
        elm_ini_get_int(ini, "pcm_matching", "icp_method", &reg->icp_method);
        elm_ini_get_int(ini, "pcm_matching", "voxel_search_method", &reg->voxel_search_method);
        elm_ini_get_double(ini, "pcm_matching", "gicp_cov_search_dist", &reg->gicp_cov_search_dist);
        elm_ini_get_int(ini, "pcm_matching", "max_thread", &reg->i_max_thread);
        elm_ini_get_int(ini, "pcm_matching", "max_iteration", &reg->max_iteration);
        elm_ini_get_double(ini, "pcm_matching", "max_search_dist", &reg->max_search_dist);
        elm_ini_get_double(ini, "pcm_matching", "lm_lambda", &reg->lm_lambda);
        elm_ini_get_double(ini, "pcm_matching", "icp_termination_threshold_m", &reg->icp_termination_threshold_m);
        elm_ini_get_double(ini, "pcm_matching", "min_overlap_ratio", &reg->min_overlap_ratio);
        elm_ini_get_double(ini, "pcm_matching", "max_fitness_score", &reg->max_fitness_score);
        elm_ini_get_bool(ini, "pcm_matching", "use_radar_cov", &reg->use_radar_cov);
        elm_ini_get_double(ini, "pcm_matching", "doppler_trans_lambda", &reg->doppler_trans_lambda);
        elm_ini_get_double(ini, "pcm_matching", "range_variance_m", &reg->range_variance_m);
        elm_ini_get_double(ini, "pcm_matching", "azimuth_variance_deg", &reg->azimuth_variance_deg);
        elm_ini_get_double(ini, "pcm_matching", "elevation_variance_deg", &reg->elevation_variance_deg);
        elm_ini_destroy(ini);
    }
    return ELM_OK;
}

extern "C" int elm_load_ekf_config(const char* localization_ini, elm_ekf_config* c) { // ekfl.cpp:250-316
    if (!localization_ini || !c) return ELM_ERR_INVALID;
    elm_ini* ini = nullptr;
    int st = elm_ini_load(localization_ini, &ini);
    if (st != ELM_OK) return st;
    const char* S = "ekf_localization";
    elm_ini_get_int(ini, S, "gps_type", &c->gps_type);
    elm_ini_get_double(ini, S, "imu_gravity", &c->imu_gravity);
    elm_ini_get_bool(ini, S, "imu_estimate_gravity", &c->imu_estimate_gravity);
    elm_ini_get_bool(ini, S, "imu_estimate_calibration", &c->imu_estimate_calibration);
    elm_ini_get_bool(ini, S, "use_zupt", &c->use_zupt);
    elm_ini_get_bool(ini, S, "use_complementary_filter", &c->use_complementary_filter);
    const std::pair<const char*, double*> keys[] = {
        {"ekf_init_x_m", &c->ekf_init_x_m}, {"ekf_init_y_m", &c->ekf_init_y_m}, {"ekf_init_z_m", &c->ekf_init_z_m},
        {"ekf_init_roll_deg", &c->ekf_init_roll_deg}, {"ekf_init_pitch_deg", &c->ekf_init_pitch_deg}, {"ekf_init_yaw_deg", &c->ekf_init_yaw_deg},
        {"ekf_state_uncertainty_pos_m", &c->state_std_pos_m}, {"ekf_state_uncertainty_rot_deg", &c->state_std_rot_deg},
        {"ekf_state_uncertainty_vel_mps", &c->state_std_vel_mps}, {"ekf_state_uncertainty_gyro_dps", &c->state_std_gyro_dps},
        {"ekf_state_uncertainty_acc_mps", &c->state_std_acc_mps}, {"ekf_imu_uncertainty_gyro_dps", &c->imu_std_gyro_dps},
        {"ekf_imu_uncertainty_acc_mps", &c->imu_std_acc_mps}, {"ekf_imu_bias_cov_gyro", &c->ekf_imu_bias_cov_gyro},
        {"ekf_imu_bias_cov_acc", &c->ekf_imu_bias_cov_acc}, {"ekf_gnss_min_cov_x_m", &c->gnss_min_cov_x_m},
        {"ekf_gnss_min_cov_y_m", &c->gnss_min_cov_y_m}, {"ekf_gnss_min_cov_z_m", &c->gnss_min_cov_z_m},
        {"ekf_gnss_min_cov_roll_deg", &c->gnss_min_cov_roll_deg}, {"ekf_gnss_min_cov_pitch_deg", &c->gnss_min_cov_pitch_deg},
        {"ekf_gnss_min_cov_yaw_deg", &c->gnss_min_cov_yaw_deg}, {"can_vel_scale_factor", &c->can_vel_scale_factor},
        {"ekf_can_meas_uncertainty_vel_mps", &c->ekf_can_meas_uncertainty_vel_mps},
        {"ekf_can_meas_uncertainty_yaw_rate_deg", &c->ekf_can_meas_uncertainty_yaw_rate_deg}};
    for (const auto& k : keys) elm_ini_get_double(ini, S, k.first, k.second);
    elm_ini_destroy(ini);
    return ELM_OK;
}

// ------------------------------------------------------------------ PCD ------------------------------------------------
namespace {
struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; size_t offset = 0; };
struct PcdHeader {
    std::vector<PcdField> fields;
    size_t width = 0, height = 1, points = 0, point_step = 0, data_pos = 0;
    bool have_points = false;
    int data = -1; // 0 ascii, 1 binary, 2 binary_compressed
};
std::vector<std::string> split_ws(const std::string& s) {
    std::vector<std::string> t;
    size_t p = 0;
    while (p < s.size()) {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\r')) ++p;
        size_t q = p;
        while (q < s.size() && !(s[q] == ' ' || s[q] == '\t' || s[q] == '\r')) ++q;
        if (q > p) t.push_back(s.substr(p, q - p));
        p = q;
    }
    return t;
}
int parse_pcd_header(const std::string& buf, PcdHeader* h) {
    size_t pos = 0;
    while (pos < buf.size()) {
        size_t eol = buf.find('\n', pos);
        if (eol == std::string::npos) eol = buf.size();
        const std::string line = buf.substr(pos, eol - pos);
        pos = eol + 1;
        const auto tok = split_ws(line);
        if (tok.empty() || tok[0][0] == '#') continue;
        const std::string& k = tok[0];
        if (k == "VERSION" || k == "VIEWPOINT") continue;
        if (k == "FIELDS" || k == "COLUMNS") {
            h->fields.assign(tok.size() - 1, PcdField());
            for (size_t i = 1; i < tok.size(); ++i) h->fields[i - 1].name = tok[i];
        } else if (k == "SIZE") {
            if (tok.size() - 1 != h->fields.size()) return ELM_ERR_INVALID;
            for (size_t i = 1; i < tok.size(); ++i) h->fields[i - 1].size = atoi(tok[i].c_str());
        } else if (k == "TYPE") {
            if (tok.size() - 1 != h->fields.size()) return ELM_ERR_INVALID;
            for (size_t i = 1; i < tok.size(); ++i) h->fields[i - 1].type = tok[i][0];
        } else if (k == "COUNT") {
            if (tok.size() - 1 != h->fields.size()) return ELM_ERR_INVALID;
            for (size_t i = 1; i < tok.size(); ++i) h->fields[i - 1].count = atoi(tok[i].c_str());
        } else if (k == "WIDTH" && tok.size() > 1) h->width = strtoull(tok[1].c_str(), nullptr, 10);
        else if (k == "HEIGHT" && tok.size() > 1) h->height = strtoull(tok[1].c_str(), nullptr, 10);
        else if (k == "POINTS" && tok.size() > 1) { h->points = strtoull(tok[1].c_str(), nullptr, 10); h->have_points = true; }
        else if (k == "DATA" && tok.size() > 1) {
            if (tok[1] == "ascii") h->data = 0;
            else if (tok[1] == "binary") h->data = 1;
            else if (tok[1] == "binary_compressed") h->data = 2;
            else return ELM_ERR_INVALID;
            h->data_pos = pos;
            break;
        }
    }
    if (h->data < 0 || h->fields.empty()) return ELM_ERR_INVALID;
    if (!h->have_points) h->points = h->width * h->height;
    size_t off = 0;
    for (auto& f : h->fields) {
        if (f.size <= 0 || f.count < 0) return ELM_ERR_INVALID;
        f.offset = off;
        off += (size_t)f.size * (size_t)f.count;
    }
    h->point_step = off;
    return ELM_OK;
}
// LZF (the codec PCL's binary_compressed uses): literal runs and back-references
bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        const unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            const size_t run = ctrl + 1;
            if (ip + run > in_len || op + run > out_len) return false;
            memcpy(out + op, in + ip, run);
            ip += run; op += run;
        } else {
            size_t len = ctrl >> 5;
            if (len == 7) { if (ip >= in_len) return false; len += in[ip++]; }
            if (ip >= in_len) return false;
            const size_t back = ((size_t)(ctrl & 0x1f) << 8) + in[ip++] + 1;
            len += 2;
            if (back > op || op + len > out_len) return false;
            for (size_t i = 0; i < len; ++i, ++op) out[op] = out[op - back]; // may overlap
        }
    }
    return op == out_len;
}
} // namespace

extern "C" void elm_free(void* p) { free(p); }

extern "C" int elm_pcd_load_xyz(const char* path, float** xyz_out, size_t* n_out) {
    if (!path || !xyz_out || !n_out) return ELM_ERR_INVALID;
    *xyz_out = nullptr;
    *n_out = 0;
    std::string buf;
    if (!read_file(path, &buf)) return ELM_ERR_IO;
    PcdHeader h;
    int st = parse_pcd_header(buf, &h);
    if (st != ELM_OK) return st;
    const PcdField* fx[3] = {nullptr, nullptr, nullptr};
    int col[3] = {-1, -1, -1};
    int c = 0;
    for (const auto& f : h.fields) {
        for (int a = 0; a < 3; ++a)
            if (f.name == (a == 0 ? "x" : a == 1 ? "y" : "z")) { fx[a] = &f; col[a] = c; }
        c += f.count;
    }
    // loadPCDFile<PointT> matches fields by name AND datatype; a map whose x/y/z are not float32 would silently load as
    // zeros there -- refuse it here instead
    for (int a = 0; a < 3; ++a)
        if (!fx[a] || fx[a]->type != 'F' || fx[a]->size != 4 || fx[a]->count != 1) return ELM_ERR_INVALID;
    const size_t n = h.points;
    // header values are untrusted: no product below may wrap, and a binary body must really hold n records
    if (h.point_step == 0 || h.point_step > (1u << 20) || (h.data != 2 && n > buf.size()) || n > (SIZE_MAX / 16) / (h.point_step > 12 ? h.point_step : 12))
        return ELM_ERR_INVALID; // (every ascii / binary record takes at least one byte of the file; a compressed body is checked below)
    for (int a = 0; a < 3; ++a)
        if ((size_t)fx[a]->offset + 4 > h.point_step) return ELM_ERR_INVALID;
    uint32_t comp = 0, uncomp = 0;
    if (h.data == 2) { // the compressed body states its own sizes: checked against the header BEFORE anything of n points is allocated
        if (buf.size() < h.data_pos + 8) return ELM_ERR_INVALID;
        memcpy(&comp, buf.data() + h.data_pos, 4);
        memcpy(&uncomp, buf.data() + h.data_pos + 4, 4);
        if (buf.size() - h.data_pos - 8 < (size_t)comp || (size_t)uncomp != n * h.point_step) return ELM_ERR_INVALID;
    }
    float* xyz = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
    if (!xyz) return ELM_ERR_ALLOC;
    if (h.data == 0) {
        size_t pos = h.data_pos, i = 0;
        while (i < n && pos < buf.size()) {
            size_t eol = buf.find('\n', pos);
            if (eol == std::string::npos) eol = buf.size();
            const auto tok = split_ws(buf.substr(pos, eol - pos));
            pos = eol + 1;
            if (tok.empty() || tok[0][0] == '#') continue;
            if ((int)tok.size() < c) { free(xyz); return ELM_ERR_INVALID; }
            for (int a = 0; a < 3; ++a) xyz[i * 3 + a] = strtof(tok[col[a]].c_str(), nullptr); // "nan" -> NaN like PCL
            ++i;
        }
        if (i != n) { free(xyz); return ELM_ERR_INVALID; }
    } else if (h.data == 1) {
        if (buf.size() < h.data_pos || buf.size() - h.data_pos < n * h.point_step) { free(xyz); return ELM_ERR_INVALID; }
        const unsigned char* d = (const unsigned char*)buf.data() + h.data_pos;
        for (size_t i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) memcpy(&xyz[i * 3 + a], d + i * h.point_step + fx[a]->offset, 4);
    } else {
        std::vector<unsigned char> raw(uncomp ? uncomp : 1);
        if (!lzf_decompress((const unsigned char*)buf.data() + h.data_pos + 8, comp, raw.data(), uncomp)) { free(xyz); return ELM_ERR_INVALID; }
        // structure-of-arrays inside: field f occupies bytes [offset_f * n, (offset_f + size_f*count_f) * n)
        for (int a = 0; a < 3; ++a) {
            const unsigned char* col_data = raw.data() + fx[a]->offset * n;
            for (size_t i = 0; i < n; ++i) memcpy(&xyz[i * 3 + a], col_data + i * 4, 4);
        }
    }
    *xyz_out = xyz;
    *n_out = n;
    return ELM_OK;
}

// ------------------------------------------------------------------ scan records ---------------------------------------
// pcl::fromROSMsg / moveFromROSMsg semantics for the two point types of pcm.hpp:81-106: fields are matched by name and
// datatype; a missing optional field leaves the member at 0.
extern "C" int elm_scan_from_cloud(const void* data, size_t n_points, size_t point_step, const elm_cloud_field* fields, int n_fields,
                                   int is_ouster, int index_sampling, float* xyz, float* intensity, float* rel_time, size_t cap, size_t* n_out) {
    if ((!data && n_points) || !fields || !xyz || !n_out) return ELM_ERR_INVALID;
    const elm_cloud_field *fxp = nullptr, *fyp = nullptr, *fzp = nullptr, *fi = nullptr, *ft = nullptr;
    const char* iname = is_ouster ? "reflectivity" : "intensity"; // dst.intensity = src.reflectivity (pcm.cpp:916)
    const char* tname = is_ouster ? "t" : "time";
    const int itype = is_ouster ? ELM_FIELD_UINT16 : ELM_FIELD_FLOAT32, ttype = is_ouster ? ELM_FIELD_UINT32 : ELM_FIELD_FLOAT32;
    for (int i = 0; i < n_fields; ++i) {
        const elm_cloud_field* f = &fields[i];
        if (!strcmp(f->name, "x") && f->datatype == ELM_FIELD_FLOAT32) fxp = f;
        else if (!strcmp(f->name, "y") && f->datatype == ELM_FIELD_FLOAT32) fyp = f;
        else if (!strcmp(f->name, "z") && f->datatype == ELM_FIELD_FLOAT32) fzp = f;
        else if (!strcmp(f->name, iname) && f->datatype == itype) fi = f;
        else if (!strcmp(f->name, tname) && f->datatype == ttype) ft = f;
    }
    if (!fxp || !fyp || !fzp) return ELM_ERR_INVALID;
    const size_t sz[5] = {4, 4, 4, is_ouster ? 2u : 4u, 4};
    const elm_cloud_field* all[5] = {fxp, fyp, fzp, fi, ft};
    for (int i = 0; i < 5; ++i)
        if (all[i] && all[i]->offset + sz[i] > point_step) return ELM_ERR_INVALID;
    const unsigned char* d = (const unsigned char*)data;
    size_t n = 0;
    if (!is_ouster) { // Cloudmsg2cloud (pcm.cpp:926-930): no index sampling
        if (cap < n_points) return ELM_ERR_INVALID;
        for (size_t i = 0; i < n_points; ++i, ++n) {
            const unsigned char* p = d + i * point_step;
            memcpy(&xyz[n * 3 + 0], p + fxp->offset, 4); memcpy(&xyz[n * 3 + 1], p + fyp->offset, 4); memcpy(&xyz[n * 3 + 2], p + fzp->offset, 4);
            float v = 0.f;
            if (fi) memcpy(&v, p + fi->offset, 4);
            if (intensity) intensity[n] = v;
            v = 0.f;
            if (ft) memcpy(&v, p + ft->offset, 4);
            if (rel_time) rel_time[n] = v;
        }
    } else { // OusterCloudmsg2cloud (pcm.cpp:900-924)
        if (index_sampling <= 0) return ELM_ERR_INVALID;
        const size_t total = (size_t)((int)(n_points / (size_t)index_sampling)) + 1; // resize(size/sampling + 1)
        if (cap < total) return ELM_ERR_INVALID;
        for (size_t i = 0; i < n_points; i += (size_t)index_sampling, ++n) {
            const unsigned char* p = d + i * point_step;
            memcpy(&xyz[n * 3 + 0], p + fxp->offset, 4); memcpy(&xyz[n * 3 + 1], p + fyp->offset, 4); memcpy(&xyz[n * 3 + 2], p + fzp->offset, 4);
            uint16_t refl = 0;
            if (fi) memcpy(&refl, p + fi->offset, 2);
            if (intensity) intensity[n] = (float)refl;
            uint32_t t = 0;
            if (ft) memcpy(&t, p + ft->offset, 4);
            if (rel_time) rel_time[n] = (float)t * 1e-9f;
        }
        for (; n < total; ++n) { // the slot the resize leaves untouched when n_points % sampling == 0: a default point
            xyz[n * 3 + 0] = xyz[n * 3 + 1] = xyz[n * 3 + 2] = 0.f;
            if (intensity) intensity[n] = 0.f;
            if (rel_time) rel_time[n] = 0.f;
        }
    }
    *n_out = n;
    return ELM_OK;
}
