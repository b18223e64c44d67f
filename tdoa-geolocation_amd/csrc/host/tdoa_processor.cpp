// tdoa_processor -- host harness over the C ABI (include/tdoa_mi355x.h) that keeps the
// reference processor's command line and file conventions (processor.go:1047-1075):
//
//   tdoa_processor [options] <ref_freq_hz> <target_freq_hz> <csv_file> <dat_file1> <dat_file2> <dat_file3> ...
//
// Same inputs: lat-lon-table.csv (Name,Latitude,Longitude,Elevation; the reference
// transmitter is the row named "%.0f" of <ref_freq_hz>, processor.go:59-99), raw .dat
// captures named after their station (substring match, processor.go:110-122), three equal
// blocks [f1 | f2 | f1] (processor.go:211-233).  Same flow: pairs i<j, reference blocks then
// target blocks, delay -> seconds -> metres, 3-station solve on the target differences.
// Default mode calls the drop-in tdoa_cross_correlate_c64 (the reference's executed chain);
// --fm runs the batched north-star path (u8 -> FM discriminator -> FFT xcorr -> peak).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/tdoa_mi355x.h"

namespace {

struct Station {
    std::string name;
    double lat = 0, lon = 0, elev = 0;
};

struct Capture {
    Station st;
    std::string path;
    std::vector<uint8_t> raw;
};

const double kC = 299792458.0;   // processor.go:873

std::string basename_of(const std::string &p)
{
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? p : p.substr(s + 1);
}

// processor.go:52-107 loadStations
bool load_stations(const std::string &csv, std::vector<Station> *out, std::string *err)
{
    std::ifstream f(csv);
    if (!f) { *err = "failed to open CSV file: " + csv; return false; }
    std::string line;
    bool header = true;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (header) { header = false; continue; }          // Name,Latitude,Longitude,Elevation
        std::stringstream ss(line);
        std::string name, a, b, c;
        if (!std::getline(ss, name, ',') || !std::getline(ss, a, ',') || !std::getline(ss, b, ',') || !std::getline(ss, c, ','))
            continue;                                         // short rows are skipped like the reference
        Station s;
        s.name = name;
        char *e1, *e2, *e3;
        s.lat = std::strtod(a.c_str(), &e1);
        s.lon = std::strtod(b.c_str(), &e2);
        s.elev = std::strtod(c.c_str(), &e3);
        if (e1 == a.c_str() || e2 == b.c_str() || e3 == c.c_str()) { *err = "invalid number in CSV row: " + line; return false; }
        out->push_back(s);
    }
    return true;
}

// >1 GiB safe whole-file read (a single os.File.Read silently stops at 1 GiB in the reference)
bool read_file(const std::string &path, std::vector<uint8_t> *out, std::string *err)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { *err = "failed to open file: " + path; return false; }
    if (std::fseek(f, 0, SEEK_END) != 0) { std::fclose(f); *err = "failed to get file size: " + path; return false; }
    const long long size = std::ftell(f);                     // -1 on a directory, FIFO or other unseekable path
    if (size < 0 || std::fseek(f, 0, SEEK_SET) != 0) { std::fclose(f); *err = "failed to get file size: " + path; return false; }
    out->resize((size_t)size);
    size_t got = 0;
    while (got < (size_t)size) {
        size_t n = std::fread(out->data() + got, 1, std::min<size_t>((size_t)size - got, 1u << 28), f);
        if (n == 0) break;
        got += n;
    }
    std::fclose(f);
    if (got != (size_t)size) { *err = "failed to read data: " + path; return false; }
    return true;
}

void usage(const char *argv0)
{
    std::printf("Usage: %s [--fm] [--fine] [--gate SAMPLES] [--device N] [--window SAMPLES] [--max-lag SAMPLES] [--k1-smooth SAMPLES] [--k1-gate] "
                "<ref_freq_hz> <target_freq_hz> <csv_file> <dat_file1> [dat_file2] [dat_file3] ...\n", argv0);
    std::printf("Example: %s 162400000 101700000 lat-lon-table.csv kx0u-data.dat n3pay-data.dat kf0mtl-data.dat\n", argv0);
}

double median(std::vector<double> v)
{
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

}  // namespace

int main(int argc, char **argv)
{
    bool fm = false, fine = false;
    double gate = 120.0;     // samples; PROJECT_NOTES.md:29-32 (max |TDOA| about 57 us = 114 samples at 2 Msps)
    tdoa_params prm;
    tdoa_default_params(&prm);
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--fm") fm = true;
        else if (a == "--fine") fm = fine = true;              // sub-sample refinement + plausibility gate (implies --fm)
        else if (a == "--gate" && i + 1 < argc) gate = std::atof(argv[++i]);
        else if (a == "--device" && i + 1 < argc) prm.device = std::atoi(argv[++i]);
        else if (a == "--window" && i + 1 < argc) prm.window_len = std::atoll(argv[++i]);
        else if (a == "--max-lag" && i + 1 < argc) prm.max_lag = std::atoi(argv[++i]);
        else if (a == "--k1-smooth" && i + 1 < argc) prm.k1_smooth = std::atoi(argv[++i]);   // the prebuilt binary's chain uses 10
        else if (a == "--k1-gate") prm.k1_gate = 1;                                          // its power gate (envelope branch)
        else pos.push_back(a);
    }
    if (pos.size() < 4) {                                     // processor.go:1048-1052
        usage(argv[0]);
        return 1;
    }
    char *end = nullptr;
    const double ref_freq = std::strtod(pos[0].c_str(), &end);
    if (end == pos[0].c_str()) { std::fprintf(stderr, "Invalid reference frequency: %s\n", pos[0].c_str()); return 1; }
    const double tgt_freq = std::strtod(pos[1].c_str(), &end);
    if (end == pos[1].c_str()) { std::fprintf(stderr, "Invalid target frequency: %s\n", pos[1].c_str()); return 1; }
    std::vector<Station> stations;
    std::string err;
    if (!load_stations(pos[2], &stations, &err)) { std::fprintf(stderr, "Failed to create processor: %s\n", err.c_str()); return 1; }
    char refname[64];
    std::snprintf(refname, sizeof(refname), "%.0f", ref_freq);          // processor.go:96
    const Station *ref = nullptr;
    for (auto &s : stations)
        if (s.name == refname) ref = &s;
    if (!ref) { std::fprintf(stderr, "Failed to create processor: reference frequency %s not found in stations\n", refname); return 1; }
    std::printf("Loaded %zu stations including reference %.0f MHz\n", stations.size(), ref_freq / 1e6);
    if (pos.size() - 3 < 3) {                                 // processor.go:740-742
        std::fprintf(stderr, "TDOA processing failed: need at least 3 collector stations, got %zu\n", pos.size() - 3);
        return 1;
    }
    std::printf("Processing TDOA for target frequency %.3f MHz\n", tgt_freq / 1e6);

    std::vector<Capture> caps;
    for (size_t i = 3; i < pos.size(); i++) {
        Capture c;
        c.path = pos[i];
        const std::string base = basename_of(pos[i]);
        bool found = false;
        for (auto &s : stations)
            if (base.find(s.name) != std::string::npos) { c.st = s; found = true; break; }
        if (!found) { std::fprintf(stderr, "TDOA processing failed: could not identify station from filename: %s\n", pos[i].c_str()); return 1; }
        size_t n_samples = 0;
        if (fm) {
            // the batched path streams the file straight to the GPU later: only its size is needed here
            FILE *f = std::fopen(pos[i].c_str(), "rb");
            if (!f) { std::fprintf(stderr, "TDOA processing failed: failed to open file: %s\n", pos[i].c_str()); return 1; }
            std::fseek(f, 0, SEEK_END);
            const long long size = std::ftell(f);
            std::fclose(f);
            if (size < 0) { std::fprintf(stderr, "TDOA processing failed: failed to get file size: %s\n", pos[i].c_str()); return 1; }
            n_samples = (size_t)size / 2;
        } else {
            if (!read_file(pos[i], &c.raw, &err)) { std::fprintf(stderr, "TDOA processing failed: %s\n", err.c_str()); return 1; }
            n_samples = c.raw.size() / 2;
        }
        std::printf("Loaded collector: %s at %.6f°, %.6f°, %.1fm (%zu samples)\n", c.st.name.c_str(), c.st.lat, c.st.lon,
                    c.st.elev, n_samples);
        caps.push_back(std::move(c));
    }
    const int S = (int)caps.size();

    std::printf("\nBaseline distances (3D):\n");                // processor.go:802-809
    for (int i = 0; i < S; i++)
        for (int j = i + 1; j < S; j++) {
            double a[3], b[3];
            tdoa_latlon_to_ecef(caps[i].st.lat, caps[i].st.lon, caps[i].st.elev, a);
            tdoa_latlon_to_ecef(caps[j].st.lat, caps[j].st.lon, caps[j].st.elev, b);
            const double d = std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
            std::printf("%s - %s: %.2f km\n", caps[i].st.name.c_str(), caps[j].st.name.c_str(), d / 1000);
        }

    tdoa_ctx *ctx = nullptr;
    int rc = tdoa_create(&prm, &ctx);
    if (rc != TDOA_OK) { std::fprintf(stderr, "tdoa_create: %s\n", tdoa_strerror(rc)); return 2; }
    auto die = [&](const char *what, int code) {
        std::fprintf(stderr, "%s: %s (%s)\n", what, tdoa_strerror(code), tdoa_last_error(ctx));
        tdoa_destroy(ctx);
        return 2;
    };

    std::vector<double> tgt_dt;                                // seconds, pairs i<j
    std::vector<double> tgt_w;
    if (!fm) {
        // ---- the reference's executed flow: complex64 signals, crossCorrelate per pair
        std::vector<std::vector<float>> refs(S), tgts(S);
        for (int s = 0; s < S; s++) {
            const size_t n = caps[s].raw.size() / 2;
            std::vector<float> data(2 * n);
            if ((rc = tdoa_load_iq_u8(ctx, caps[s].raw.data(), n, data.data()))) return die("tdoa_load_iq_u8", rc);
            const size_t bs = n / 3;                           // processor.go:214
            if (bs == 0) { refs[s] = data; tgts[s] = data; continue; }
            refs[s].assign(data.begin(), data.begin() + 2 * bs);                                   // block 1
            refs[s].insert(refs[s].end(), data.begin() + 4 * bs, data.begin() + 6 * bs);           // block 3
            tgts[s].assign(data.begin() + 2 * bs, data.begin() + 4 * bs);                          // block 2
            const size_t chunk = (size_t)prm.window_len;       // processor.go:772-780
            if (refs[s].size() / 2 > chunk) refs[s].resize(2 * chunk);
            if (tgts[s].size() / 2 > chunk) tgts[s].resize(2 * chunk);
        }
        for (int kind = 0; kind < 2; kind++) {
            std::printf(kind == 0 ? "\n=== REFERENCE SIGNAL CORRELATION TEST ===\n" : "\n=== TARGET SIGNAL CORRELATION TEST ===\n");
            auto &sig = kind == 0 ? refs : tgts;
            // one call for all pairs: every station's signal goes through preprocessSignal once instead of once per pair
            // (processor.go:629-630); the numbers are the per-pair crossCorrelate's, bit for bit
            std::vector<const float *> ptr(S);
            std::vector<size_t> len(S);
            for (int s = 0; s < S; s++) { ptr[s] = sig[s].data(); len[s] = sig[s].size() / 2; }
            std::vector<int32_t> delays((size_t)S * (S - 1) / 2);
            std::vector<double> corrs(delays.size());
            if ((rc = tdoa_cross_correlate_batch_c64(ctx, ptr.data(), len.data(), S, delays.data(), corrs.data())))
                return die("tdoa_cross_correlate_batch_c64", rc);
            int pidx = 0;
            for (int i = 0; i < S; i++)
                for (int j = i + 1; j < S; j++, pidx++) {
                    const int32_t delay = delays[pidx];
                    const double corr = corrs[pidx];
                    const double dt = (double)delay / prm.sample_rate;       // processor.go:821-822
                    std::printf("%s %s - %s: delay=%d samples (%.3f μs), correlation=%.17g\n", kind == 0 ? "REF" : "TGT",
                                caps[i].st.name.c_str(), caps[j].st.name.c_str(), delay, dt * 1e6, corr);
                    if (kind == 1) { tgt_dt.push_back(dt); tgt_w.push_back(std::fabs(corr)); }
                }
        }
    } else {
        // ---- north-star path: raw bytes in, one peak per (window, pair) out
        for (int s = 0; s < S; s++)   // file -> pinned staging -> HBM
            if ((rc = tdoa_capture_upload_file(ctx, s, caps[s].path.c_str(), nullptr))) return die("tdoa_capture_upload_file", rc);
        int wpb = 0, W = 0;
        if ((rc = tdoa_num_windows(ctx, &wpb, &W))) return die("tdoa_num_windows", rc);
        const int P = tdoa_num_pairs(ctx);
        {
            // capture QA in the spirit of collector.go:204-248 (validateDataFile), on every window instead of 1000 samples:
            // mean block power per station, REF blocks consistent within 2x, TGT block different from REF by > 50 %
            std::vector<tdoa_window_quality> q((size_t)W * S);
            if ((rc = tdoa_window_quality_all(ctx, 0, 1, q.data()))) return die("tdoa_window_quality_all", rc);
            std::printf("\n=== CAPTURE QUALITY (mean power of (I-127.5)^2+(Q-127.5)^2 per block) ===\n");
            for (int s = 0; s < S; s++) {
                double pw[3] = {0, 0, 0};
                int clip = 0, overload = 0;
                for (int w = 0; w < W; w++) {
                    const tdoa_window_quality &x = q[(size_t)w * S + s];
                    pw[w / wpb] += x.mean_power / wpb;
                    clip += x.has_clipping;
                    overload += x.has_overload;
                }
                // an all-127/128 block has power 0.25 at least, but an empty or all-127.5-equivalent one must not print inf/NaN
                const auto ratio = [](double a, double b) { return b > 0 ? a / b : 0.0; };
                const double ref_ratio = ratio(pw[2], pw[0]);                       // collector.go:231
                const double tgt_ratio = (ratio(pw[1], pw[0]) + ratio(pw[1], pw[2])) / 2.0; // collector.go:240-242
                std::printf("%s: REF %.2f  TGT %.2f  REF %.2f  | REF blocks %s (%.2fx), TGT/REF %.2fx%s | %d clipped, %d low-level windows of %d\n",
                            caps[s].st.name.c_str(), pw[0], pw[1], pw[2],
                            (ref_ratio > 2.0 || ref_ratio < 0.5) ? "INCONSISTENT" : "consistent", ref_ratio, tgt_ratio,
                            (tgt_ratio > 1.5 || tgt_ratio < 0.67) ? "" : " (very similar)", clip, overload, W);
            }
        }
        std::vector<tdoa_peak> peaks((size_t)W * P);
        std::vector<tdoa_fine_peak> fines;
        if (fine) {
            fines.resize((size_t)W * P);
            if ((rc = tdoa_process_fine(ctx, 0, 1, gate, peaks.data(), fines.data()))) return die("tdoa_process_fine", rc);
        } else if ((rc = tdoa_process(ctx, 0, 1, peaks.data(), nullptr))) return die("tdoa_process", rc);
        std::printf("\n=== FM-DISCRIMINATOR CROSS-CORRELATION: %d windows x %d pairs ===\n", W, P);
        int p = 0;
        for (int i = 0; i < S; i++)
            for (int j = i + 1; j < S; j++, p++) {
                std::vector<double> lr, lt, cr, ct;
                for (int w = 0; w < W; w++) {
                    const tdoa_peak &pk = peaks[(size_t)w * P + p];
                    const bool target = (w / wpb) == 1;        // block 1 is the target frequency
                    (target ? lt : lr).push_back(pk.lag);
                    (target ? ct : cr).push_back(std::fabs(pk.corr));
                }
                std::printf("REF %s - %s: median lag=%.0f samples over %zu windows, median |corr|=%.6f\n", caps[i].st.name.c_str(),
                            caps[j].st.name.c_str(), median(lr), lr.size(), median(cr));
                std::printf("TGT %s - %s: median lag=%.0f samples over %zu windows, median |corr|=%.6f\n", caps[i].st.name.c_str(),
                            caps[j].st.name.c_str(), median(lt), lt.size(), median(ct));
                double lag_used = median(lt);
                if (fine) {
                    // refined delays of the target windows that pass the gate (all of them if none does)
                    std::vector<double> ok, all;
                    for (int w = 0; w < W; w++) {
                        if ((w / wpb) != 1) continue;
                        const tdoa_fine_peak &fk = fines[(size_t)w * P + p];
                        all.push_back(fk.delay);
                        if (fk.plausible) ok.push_back(fk.delay);
                    }
                    lag_used = median(ok.empty() ? all : ok);
                    if (ok.empty()) ct.clear();                    // no plausible window: the pair gets weight 0 in the N-station solve
                    std::printf("TGT %s - %s: refined delay=%.3f samples, %zu of %zu windows within +-%.1f samples\n",
                                caps[i].st.name.c_str(), caps[j].st.name.c_str(), lag_used, ok.size(), all.size(), gate);
                }
                tgt_dt.push_back(lag_used / prm.sample_rate);
                tgt_w.push_back(median(ct));
            }
    }

    // ---- downstream: range differences and the position solve (processor.go:892-926)
    std::vector<double> rd(tgt_dt.size());
    std::printf("\n=== TDOA GEOLOCATION ===\nTime differences (μs): ");
    for (size_t i = 0; i < tgt_dt.size(); i++) { rd[i] = tgt_dt[i] * kC; std::printf("%.3f ", tgt_dt[i] * 1e6); }
    std::printf("\nRange differences (m): ");
    for (double v : rd) std::printf("%.1f ", v);
    std::printf("\n");
    std::vector<double> lle(3 * S);
    for (int s = 0; s < S; s++) { lle[3 * s] = caps[s].st.lat; lle[3 * s + 1] = caps[s].st.lon; lle[3 * s + 2] = caps[s].st.elev; }
    double out[3];
    int iters = 0;
    if (S == 3) rc = tdoa_solve_3station(lle.data(), rd.data(), out, &iters);
    else rc = tdoa_solve_nstation(lle.data(), S, rd.data(), tgt_w.data(), 0, out, &iters);   // weights: median |corr| per pair
    if (rc != TDOA_OK) {                                       // processor.go:919-921
        if (rc == TDOA_ERR_INVALID)
            std::fprintf(stderr, "TDOA solution failed: fewer usable station pairs than unknowns (no target window of the "
                                 "other pairs passed the +-%.1f-sample plausibility gate, or a weight is not finite)\n", gate);
        else
            std::fprintf(stderr, "TDOA solution failed: %s at iteration %d\n", tdoa_strerror(rc), iters);
        tdoa_destroy(ctx);
        return 3;
    }
    std::printf("\n*** CALCULATED TRANSMITTER LOCATION ***\nLatitude:  %.6f°\nLongitude: %.6f°\nElevation: %.1f m\n", out[0], out[1], out[2]);
    tdoa_destroy(ctx);
    return 0;
}
