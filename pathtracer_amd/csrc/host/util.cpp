// util.cpp — error text, logger and flag parser of the host pipeline.
// Restates src/logger.cu (levels, "[LEVEL] msg" to file and stdout, 1 KiB line
// buffer), src/config.cu (flag names and the argv scan) and the file helper
// the loaders share.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "host_internal.h"

namespace pth {

static thread_local char g_error[1024] = "";

int set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return 1;
}

bool read_file(const std::string &path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) {
        fclose(f);
        return false;
    }
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

} // namespace pth

extern "C" const char *pth_last_error(void) { return pth::g_error; }

extern "C" void pth_free(void *p) { free(p); }

// ---- logger (src/logger.cu) ---------------------------------------------------

static int g_log_level = 3; // LOG_LEVEL_INFO, src/logger.cu:9
static FILE *g_log_file = nullptr;
static int g_log_stdout = 0; // library default is quiet; the CLI turns it on
static std::mutex g_log_mutex;

extern "C" int pth_log_init(int level, const char *file, int to_stdout) {
    std::lock_guard<std::mutex> lock(g_log_mutex);
    g_log_level = level;
    g_log_stdout = to_stdout;
    if (g_log_file) {
        fclose(g_log_file);
        g_log_file = nullptr;
    }
    if (file && file[0]) {
        g_log_file = fopen(file, "w");
        if (!g_log_file) {
            fprintf(stderr, "unable to open logfile at: %s\n", file);
            return 1;
        }
    }
    return 0;
}

extern "C" void pth_log_close(void) {
    std::lock_guard<std::mutex> lock(g_log_mutex);
    if (g_log_file) {
        fclose(g_log_file);
        g_log_file = nullptr;
    }
}

extern "C" void pth_log(int level, const char *fmt, ...) {
    if (level > g_log_level) return;
    static const char *prefix[] = {"NONE", "ERROR", "WARNING", "INFO", "TRACE"};
    char buffer[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buffer, sizeof(buffer), fmt, ap);
    va_end(ap);
    const char *p = (level >= 0 && level <= 4) ? prefix[level] : "LOG";
    std::lock_guard<std::mutex> lock(g_log_mutex);
    if (g_log_file) {
        fprintf(g_log_file, "[%s] %s", p, buffer);
        fflush(g_log_file);
    }
    if (g_log_stdout) {
        fprintf(stdout, "[%s] %s", p, buffer);
        fflush(stdout);
    }
}

// ---- flags (src/config.cu) ------------------------------------------------------

namespace {

typedef int (*tokenizer_fn)(void *field, const char *input);

int tok_vec3(void *field, const char *input) {
    pt_vec3 *v = (pt_vec3 *)field;
    if (3 != sscanf(input, "%f %f %f", &v->x, &v->y, &v->z)) {
        fprintf(stderr, "unable to parse vec3 arg input=%s\n", input);
        return 1;
    }
    return 0;
}
int tok_int(void *field, const char *input) {
    if (1 != sscanf(input, "%d", (int *)field)) {
        fprintf(stderr, "unable to parse int arg input=%s\n", input);
        return 1;
    }
    return 0;
}
int tok_float(void *field, const char *input) {
    if (1 != sscanf(input, "%f", (float *)field)) {
        fprintf(stderr, "unable to parse float arg input=%s\n", input);
        return 1;
    }
    return 0;
}
int tok_path(void *field, const char *input) {
    strncpy((char *)field, input, PTH_PATH_MAX - 1);
    ((char *)field)[PTH_PATH_MAX - 1] = '\0';
    return 0;
}

// src/config.cu:38-53 — the flag must sit at argv[1..argc-2]; its value is the
// next argument; a tokenizer failure is ignored.
int find_flag(int argc, char **argv, void *field, const char *name, tokenizer_fn tok, bool optional) {
    for (int i = 1; i < argc - 1; i++) {
        if (!strcmp(argv[i], name)) {
            tok(field, argv[i + 1]);
            return 0;
        }
    }
    if (!optional) {
        fprintf(stderr, "ERROR unable to find required config item '%s'.\n", name);
        return 1;
    }
    return 0;
}

} // namespace

extern "C" int pth_load_config(pth_config *cfg, int argc, char **argv) {
    memset(cfg, 0, sizeof(*cfg));
    int missing = 0;
    missing += find_flag(argc, argv, &cfg->world_clear_color, "--world-clear-color", tok_vec3, true);
    missing += find_flag(argc, argv, cfg->world_clear_color_texture, "--world-clear-color-texture", tok_path, false);
    missing += find_flag(argc, argv, &cfg->seed, "--sampling-seed", tok_int, false);
    missing += find_flag(argc, argv, &cfg->samples, "--sampling-samples", tok_int, false);
    missing += find_flag(argc, argv, &cfg->samples_every_update, "--sampling-samples-every-update", tok_int, false);
    missing += find_flag(argc, argv, &cfg->resolution_x, "--output-resolution-x", tok_int, false);
    missing += find_flag(argc, argv, &cfg->resolution_y, "--output-resolution-y", tok_int, false);
    missing += find_flag(argc, argv, &cfg->output_exposure, "--output-exposure", tok_float, false);
    missing += find_flag(argc, argv, &cfg->log_level, "--logger-log-level", tok_int, false);
    missing += find_flag(argc, argv, &cfg->log_stdout, "--logger-log-stdout", tok_int, false);
    missing += find_flag(argc, argv, cfg->path_gltf, "--path-gltf", tok_path, false);
    missing += find_flag(argc, argv, cfg->dir_output, "--dir-output", tok_path, false);
    missing += find_flag(argc, argv, &cfg->default_camera_position, "--default-camera-position", tok_vec3, false);
    missing += find_flag(argc, argv, &cfg->default_camera_target, "--default-camera-target", tok_vec3, false);
    missing += find_flag(argc, argv, &cfg->default_camera_updir, "--default-camera-updir", tok_vec3, false);
    missing += find_flag(argc, argv, &cfg->default_camera_yfov, "--default-camera-yfov", tok_float, false);
    return missing;
}
