// main.cpp — bin/pathtracer: the reference's CLI (src/main.cu:36-156) on top of the C ABIs.
//
// Same flags (src/config.cu:63-83), same outputs (<dir-output>/log.txt and render.png,
// rewritten after every batch), same batch loop (seed + 1 per batch, running mean) and the
// same "Rendered a out of b S/px in t s - x S/px/s - y MS/s" line, whose clock — like the
// reference's — includes the device->host copy and the PNG encode of every batch
// (main.cu:98,116-131). The kernel-only rate is logged next to it. Exit-on-error policy as in
// the reference (utils.cu:16-22).
#include <sys/stat.h>
#include <sys/types.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ptcore.h"
#include "pthost.h"

static void human_readable_size(char *buf, size_t n, unsigned long long bytes) { // utils.cu:24-35
    if (bytes < 1000ull) snprintf(buf, n, "%lluB", bytes);
    else if (bytes < 1000000ull) snprintf(buf, n, "%lluKB", bytes / 1000ull);
    else if (bytes < 1000000000ull) snprintf(buf, n, "%lluMB", bytes / 1000000ull);
    else snprintf(buf, n, "%lluGB", bytes / 1000000000ull);
}

#define CORE_TRY(call)                                              \
    do {                                                            \
        if (call) {                                                 \
            fprintf(stderr, "HIP error: %s\n", pt_last_error());    \
            pth_log(1, "%s\n", pt_last_error());                    \
            exit(EXIT_FAILURE);                                     \
        }                                                           \
    } while (0)

int main(int argc, char *argv[]) {
    printf("Reading config from args: ");
    for (int i = 0; i < argc; i++) printf("%s ", argv[i]);
    printf("\n\n");

    pth_config cfg;
    pth_load_config(&cfg, argc, argv);
    if (cfg.resolution_x <= 0 || cfg.resolution_y <= 0 || cfg.samples <= 0 || cfg.samples_every_update <= 0 ||
        !cfg.path_gltf[0] || !cfg.dir_output[0]) {
        fprintf(stderr, "missing or invalid required flags (resolution, samples, --path-gltf, --dir-output)\n");
        return 1;
    }

    struct stat st;
    if (stat(cfg.dir_output, &st) == -1) mkdir(cfg.dir_output, 0700);
    char log_file[PTH_PATH_MAX + 16];
    snprintf(log_file, sizeof(log_file), "%s/log.txt", cfg.dir_output);
    snprintf(cfg.path_render, sizeof(cfg.path_render), "%.240s/render.png", cfg.dir_output);
    if (pth_log_init(cfg.log_level, log_file, cfg.log_stdout)) return 1;

    pth_scene *scene = nullptr;
    if (pth_scene_load_gltf(cfg.path_gltf, &cfg, &scene)) {
        fprintf(stderr, "%s\n", pth_last_error());
        return EXIT_FAILURE;
    }
    const pt_scene_desc *desc = pth_scene_get_desc(scene);

    pth_log(3, "Building BVH...  \n");
    pth_bvh bvh;
    if (pth_bvh_build(desc, &bvh)) {
        pth_log(1, "%s\n", pth_last_error());
        return EXIT_FAILURE;
    }
    pth_log(3, "Done building BVH\n");

    pth_log(3, "Building LST... \n");
    pt_lst_node *lst = nullptr;
    uint32_t lst_count = 0;
    if (pth_lst_build(desc, &lst, &lst_count)) {
        pth_log(1, "%s\n", pth_last_error());
        return EXIT_FAILURE;
    }
    pth_log(3, "Done building LST\n");

    size_t pixels = (size_t)cfg.resolution_x * (size_t)cfg.resolution_y;
    std::vector<float> h_img(pixels * 3);

    // ########## DISPATCH #############
    int device = 0;
    if (const char *env = getenv("PT_DEVICE")) device = atoi(env);

    // --gpus N (not a reference flag: the reference is single-GPU): devices PT_DEVICE .. PT_DEVICE + N - 1 of this
    // node render the batches b = rank (mod N) side by side and ONE RCCL sum-reduce combines their accumulators
    // (pt_group_render). The picture is written once, at the end.
    int gpus = 0;
    for (int i = 1; i + 1 < argc; i++)
        if (strcmp(argv[i], "--gpus") == 0) gpus = atoi(argv[i + 1]);
    if (gpus > 0) {
        const int present = pt_device_count();
        if (device < 0 || device + gpus > present) {
            fprintf(stderr, "--gpus %d: this node has %d HIP device%s and PT_DEVICE (the first one to use) is %d; the ranks need devices %d..%d\n",
                    gpus, present, present == 1 ? "" : "s", device, device, device + gpus - 1);
            pth_log(1, "--gpus %d needs devices %d..%d, %d present\n", gpus, device, device + gpus - 1, present);
            return EXIT_FAILURE;
        }
        std::vector<int> devices;
        for (int i = 0; i < gpus; i++) devices.push_back(device + i);
        pt_group *group = nullptr;
        CORE_TRY(pt_group_create(devices.data(), gpus, &group));
        pt_render_config rc;
        rc.resolution_x = cfg.resolution_x;
        rc.resolution_y = cfg.resolution_y;
        rc.output_exposure = cfg.output_exposure;
        rc.world_clear_color = cfg.world_clear_color;
        pth_log(3, "Copying scene, bvh_t and lst_t to %d devices... \n", gpus);
        for (int r = 0; r < gpus; r++) {
            pt_context *c = pt_group_context(group, r);
            CORE_TRY(pt_scene_upload(c, desc));
            CORE_TRY(pt_bvh_upload(c, bvh.nodes, bvh.node_count, bvh.indices, bvh.index_count));
            CORE_TRY(pt_lst_upload(c, lst, lst_count));
            CORE_TRY(pt_configure(c, &rc));
        }
        pth_log(3, "Done\n");
        pth_log(3, "Rendering %d samples in batches of %d on %d GPUs, img size (%d, %d)\n", cfg.samples, cfg.samples_every_update,
                gpus, cfg.resolution_x, cfg.resolution_y);
        auto start = std::chrono::system_clock::now();
        CORE_TRY(pt_group_render(group, cfg.seed, cfg.samples, cfg.samples_every_update));
        float render_s = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::system_clock::now() - start).count() / 1e6f;
        CORE_TRY(pt_read_image(pt_group_context(group, 0), h_img.data()));
        if (pth_write_image(h_img.data(), cfg.resolution_x, cfg.resolution_y, cfg.path_render)) {
            pth_log(1, "%s\n", pth_last_error());
            exit(EXIT_FAILURE);
        }
        float elapsed = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - start).count() / 1000.0f;
        pth_log(3, "Rendered %d out of %d S/px in %.1fs - %.2f S/px/s - %.2f MS/s\n", cfg.samples, cfg.samples, elapsed,
                cfg.samples / elapsed, (float)cfg.resolution_x * cfg.resolution_y * cfg.samples / (1000000.0 * elapsed));
        pth_log(3, "Render + reduce only: %.3fs - %.2f MS/s on %d GPUs\n", render_s,
                (double)pixels * cfg.samples / (1e6 * (render_s > 0 ? render_s : 1e-9)), gpus);
        pt_group_destroy(group);
        pth_free(lst);
        pth_bvh_free(&bvh);
        pth_scene_free(scene);
        pth_log_close();
        return 0;
    }
    pt_context *ctx = nullptr;
    CORE_TRY(pt_create(device, &ctx));
    char buf[64];
    pth_log(3, "Copying scene to device... \n");
    CORE_TRY(pt_scene_upload(ctx, desc));
    unsigned long long after_scene = pt_device_footprint(ctx);
    human_readable_size(buf, sizeof(buf), after_scene);
    pth_log(3, "Done [%s]\n", buf);
    pth_log(3, "Copying bvh_t to device... \n");
    CORE_TRY(pt_bvh_upload(ctx, bvh.nodes, bvh.node_count, bvh.indices, bvh.index_count));
    unsigned long long after_bvh = pt_device_footprint(ctx);
    human_readable_size(buf, sizeof(buf), after_bvh - after_scene);
    pth_log(3, "Done [%s]\n", buf);
    pth_log(3, "Copying lst_t to device... \n");
    CORE_TRY(pt_lst_upload(ctx, lst, lst_count));
    human_readable_size(buf, sizeof(buf), (unsigned long long)lst_count * sizeof(pt_lst_node));
    pth_log(3, "Done [%s]\n", buf);

    pt_render_config rc;
    rc.resolution_x = cfg.resolution_x;
    rc.resolution_y = cfg.resolution_y;
    rc.output_exposure = cfg.output_exposure;
    rc.world_clear_color = cfg.world_clear_color;
    CORE_TRY(pt_configure(ctx, &rc));

    pth_log(3, "Launching kernel... \n");
    pth_log(3, "Rendering %d samples in batches of %d, img size (%d, %d)\n", cfg.samples, cfg.samples_every_update,
            cfg.resolution_x, cfg.resolution_y);
    pth_log(3, "Kernel params <<<(%d), (64)>>> (one 8x8-pixel tile per wave64 workgroup)\n",
            ((cfg.resolution_x + 7) / 8) * ((cfg.resolution_y + 7) / 8));

    // The reference copies the image back, tonemaps it and rewrites render.png after EVERY batch, inside
    // the timed loop (main.cu:108-131). Same files, same log lines here, but that host work (about as long
    // as a 50-spp launch at 1024^2) runs on a worker thread beside the next batch's kernel: the main thread
    // reads batch k into one of two host buffers, waits until the worker is done with batch k-1, hands
    // over, and launches batch k+1. Every "Rendered ..." line is still printed once its PNG is on disk.
    // The two host buffers are page-locked where the runtime grants it (the copy back then runs at the DMA engine's
    // rate); pageable memory otherwise, as in the reference (main.cu:86).
    std::vector<float> h_img2;
    float *pinned[2] = {(float *)pt_host_alloc(pixels * 3 * sizeof(float)), (float *)pt_host_alloc(pixels * 3 * sizeof(float))};
    if (!pinned[1]) h_img2.resize(pixels * 3);
    float *buffers[2] = {pinned[0] ? pinned[0] : h_img.data(), pinned[1] ? pinned[1] : h_img2.data()};
    auto start = std::chrono::system_clock::now();
    std::thread worker;
    bool worker_failed = false;
    double kernel_seconds = 0;
    int rendered = 0, batch = 0;
    while (rendered < cfg.samples) {
        int current = cfg.samples_every_update < cfg.samples - rendered ? cfg.samples_every_update : cfg.samples - rendered;
        auto t0 = std::chrono::steady_clock::now();
        CORE_TRY(pt_render_batch(ctx, cfg.seed, rendered, current));
        kernel_seconds += pt_last_kernel_ms(ctx) / 1000.0;
        auto t1 = std::chrono::steady_clock::now();
        float *buf = buffers[batch & 1];
        CORE_TRY(pt_read_image(ctx, buf));
        auto t2 = std::chrono::steady_clock::now();
        rendered += current;
        cfg.seed++;
        batch++;
        if (worker.joinable()) worker.join();
        if (worker_failed) exit(EXIT_FAILURE);
        auto t3 = std::chrono::steady_clock::now();
        pth_log(4, "Batch %d on the main thread: launch to kernel end %.2f ms (kernel %.2f), copy back %.2f ms, waited for the writer %.2f ms\n", batch,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), pt_last_kernel_ms(ctx),
                std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double, std::milli>(t3 - t2).count());
        worker = std::thread([&, buf, rendered, kernel_seconds]() {
            auto w0 = std::chrono::steady_clock::now();
            if (pth_write_image(buf, cfg.resolution_x, cfg.resolution_y, cfg.path_render)) {
                pth_log(1, "%s\n", pth_last_error());
                worker_failed = true;
                return;
            }
            pth_log(4, "Tonemap + PNG on the writer thread: %.2f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
            auto now = std::chrono::system_clock::now();
            int elapsed_ms = (int)std::chrono::duration_cast<std::chrono::milliseconds>(now - start).count();
            float elapsed = elapsed_ms / 1000.0f;
            float ms_per_s = (float)cfg.resolution_x * cfg.resolution_y * rendered / (1000000.0 * elapsed);
            float spp_per_s = rendered / elapsed;
            pth_log(3, "Rendered %d out of %d S/px in %.1fs - %.2f S/px/s - %.2f MS/s\n", rendered, cfg.samples, elapsed,
                    spp_per_s, ms_per_s);
            pth_log(3, "Kernel only: %.3fs - %.2f MS/s\n", kernel_seconds,
                    (double)pixels * rendered / (1e6 * (kernel_seconds > 0 ? kernel_seconds : 1e-9)));
            pth_log(4, "Updated image: %s\n", cfg.path_render);
        });
    }
    if (worker.joinable()) worker.join();
    if (worker_failed) exit(EXIT_FAILURE);
    // ########## DISPATCH #############

    pt_host_free(pinned[0]);
    pt_host_free(pinned[1]);
    pt_destroy(ctx);
    pth_free(lst);
    pth_bvh_free(&bvh);
    pth_scene_free(scene);
    pth_log_close();
    return 0;
}
