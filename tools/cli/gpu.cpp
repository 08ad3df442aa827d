// gpu.cpp -- ./gpu <img-path> <psf-length> <psf-angle> [--out file] [--mode fast|parity] [--norm padded|cropped] [--host-epilogue]
// Drop-in counterpart of the reference's gpu.cpp (argument meaning, printed lines and exit codes as at
// gpu.cpp:57-138 of the reference): read image, /255, PSF, K = 0.01, split BGR, warm-up call, timed
// wienerDeblur_RGB_optimized, timed wienerDeblur_RGB_naive, merge, Lab white balance, 8-bit result.
// The serial leg the original runs first (gpu.cpp:83-91) is here too, through the same names (autoPadToPowerOfTwo ->
// fft_serial::wienerDeblur_myfft -> crop): in this repository fft_serial:: runs on the GPU in the parity mode, whose
// pixels are bit-identical to ./serial (tests/), so both "[Speedup]" lines divide that leg's time by a GPU entry
// point's, as gpu.cpp:105,113 do.  There is no CPU code path in this binary.
#include "utils.hpp"
#include "fft/fft.hpp"
#include "fdr_image_io.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <string>

// gpu.cpp:13-55 of the reference: L-inf <= epsilon per channel, else PSNR >= 30 dB still passes ("floating point
// drift").  The reference compares serial vs GPU with it (call commented out at gpu.cpp:116-121); --verify runs it on
// the serial leg's planes (parity mode: the serial path's pixels) against this run's planes: epsilon 1e-4, the
// tolerance BASELINE states.  What it proves is that the two MODES of this library agree; that the parity mode equals
// ./serial is the business of tests/ (against the CPU oracle), and the message says so.
static bool areChannelsEqual(const vector<Mat>& vec1, const vector<Mat>& vec2, double epsilon = 1e-4) {
    if (vec1.size() != vec2.size()) { cerr << "Error: Channel count mismatch.\n"; return false; }
    for (size_t i = 0; i < vec1.size(); ++i) {
        const Mat &m1 = vec1[i], &m2 = vec2[i];
        if (m1.rows != m2.rows || m1.cols != m2.cols || m1.type() != m2.type()) { cerr << "Error: Size/Type mismatch.\n"; return false; }
        double diff = 0.0, sq = 0.0;
        for (int r = 0; r < m1.rows; ++r)
            for (int c = 0; c < m1.cols; ++c) {
                const double d = (double)m1.ptr<float>(r)[c] - (double)m2.ptr<float>(r)[c];
                diff = std::max(diff, std::fabs(d));
                sq += d * d;
            }
        const double mse = sq / ((double)m1.rows * m1.cols);
        const double psnr = mse > 1e-10 ? 10.0 * log10(1.0 / mse) : 100.0;
        if (diff > epsilon) {
            if (psnr >= 30.0) {
                cout << "[Info] Channel " << i << " has floating point drift."
                     << "\n       Max Diff: " << diff << "\n       PSNR: " << psnr << " dB (Excellent! > 30dB is good)"
                     << "\n       -> Verification PASSED (Relaxed)." << endl;
            } else {
                cerr << "[Error] Content mismatch in channel " << i << ".\n";
                cerr << "       Max pixel difference: " << diff << " (Threshold: " << epsilon << ")\n";
                cerr << "       PSNR: " << psnr << " dB (Too low!)\n";
                return false;
            }
        }
    }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        cout << "Usage: ./gpu <img-path> <psf-length> <psf-angle>\n";
        return -1;
    }
    string img_path = argv[1];
    int psf_length = atoi(argv[2]);
    double psf_angle = atof(argv[3]);
    string out_path, raw_path;
    bool verify = false;         // --verify: areChannelsEqual(parity-mode result, this run's result)
    bool host_epilogue = false;  // Lab white balance on the host (the A/B reference of the device epilogue)
    for (int i = 4; i < argc; ++i) {
        string a = argv[i];
        if (a == "--out" && i + 1 < argc) out_path = argv[++i];
        else if (a == "--raw-out" && i + 1 < argc) raw_path = argv[++i];  // restored float planes B,G,R before white balance
        else if (a == "--host-epilogue") host_epilogue = true;
        else if (a == "--verify") verify = true;
        else if (a == "--mode" && i + 1 < argc) fft_gpu::set_mode(string(argv[++i]) == "parity" ? FDR_MODE_PARITY : FDR_MODE_FAST);
        else if (a == "--norm" && i + 1 < argc) fft_gpu::set_norm_area(string(argv[++i]) == "cropped" ? FDR_NORM_CROPPED : FDR_NORM_PADDED);
        else { cout << "Usage: ./gpu <img-path> <psf-length> <psf-angle>\n"; return -1; }
    }

    Mat img = fdr_io::imread(img_path);
    if (img.empty()) { cout << "Cannot read image\n"; return -1; }
    img.convertTo(img, CV_32F);
    img /= 255.0;

    Mat psf = motionBlurKernel(psf_length, psf_angle);
    float K = 0.01f;

    vector<Mat> channels;
    split(img, channels);
    vector<Mat> input = channels;

    // serial leg (gpu.cpp:83-91): pad, fft_serial::wienerDeblur_myfft, crop -- prints the accumulated phase block
    // of fft/fft_serial.cpp:249-258 on its third call
    vector<Mat> serial_channels = input;
    auto t_start = high_resolution_clock::now();
    for (int i = 0; i < 3; i++) {
        const Mat padded = autoPadToPowerOfTwo(serial_channels[i]);
        const Mat restored = fft_serial::wienerDeblur_myfft(padded, psf, K);
        serial_channels[i] = restored(Rect(0, 0, img.cols, img.rows)).clone();
    }
    auto t_end = high_resolution_clock::now();
    const double serial_time = getElapsedMs(t_start, t_end);
    cout << "Deblurring 3 channels took(serial): " << serial_time << " ms\n";
    // (what that leg is HERE: fft_serial:: on the GPU in the parity mode, one plan + PSF spectrum per channel, run cold --
    // this binary has no CPU path, so the two [Speedup] ratios below compare GPU parity-mode-cold with GPU fast mode, not a
    // CPU with a GPU as gpu.cpp:105,113 of the reference do)
    cout << "[Note] serial leg = fft_serial:: names on the GPU (parity mode, cold); the speed-up lines divide that leg by a GPU entry point\n";

    fft_gpu::wienerDeblur_RGB_optimized(channels, psf, K);  // warm-up, as gpu.cpp:96 (restores in place)

    channels = input;
    t_start = high_resolution_clock::now();
    fft_gpu::wienerDeblur_RGB_optimized(channels, psf, K);
    t_end = high_resolution_clock::now();
    const double opt_time = getElapsedMs(t_start, t_end);
    cout << "Deblurring 3 channels took(gpu[optimize]): " << opt_time << " ms\n";
    printf("[Speedup] %.2fx ms\n", serial_time / opt_time);

    vector<Mat> naive = input;
    t_start = high_resolution_clock::now();
    fft_gpu::wienerDeblur_RGB_naive(naive, psf, K);
    t_end = high_resolution_clock::now();
    const double naive_time = getElapsedMs(t_start, t_end);
    cout << "Deblurring 3 channels took(gpu): " << naive_time << " ms\n";
    printf("[Speedup] %.2fx ms\n", serial_time / naive_time);

    if (verify) {  // the check of gpu.cpp:116-121 between the serial leg's planes and this run's planes
        if (areChannelsEqual(serial_channels, channels))
            cout << "[Success] fast mode matches the serial-equivalent parity mode (L-inf <= 1e-4 per channel, or PSNR >= 30 dB).\n";
        else
            cout << "[Error] fast mode and the serial-equivalent parity mode differ.\n";
    }

    if (!raw_path.empty()) {
        FILE* f = fopen(raw_path.c_str(), "wb");
        if (!f) { cout << "Cannot write " << raw_path << "\n"; return -1; }
        for (const Mat& c : channels)
            for (int r = 0; r < c.rows; ++r) fwrite(c.ptr<float>(r), sizeof(float), (size_t)c.cols, f);
        fclose(f);
    }

    Mat corrected_BGR;
    if (host_epilogue) {  // the reference's sequence on the host (gpu.cpp:123-137)
        Mat merged_float;
        merge(channels, merged_float);
        Mat merged_Lab = fdr_io::bgr2lab(merged_float), img_orig_Lab = fdr_io::bgr2lab(img);
        Mat corrected_Lab = applyWhiteBalance(merged_Lab, img_orig_Lab);
        corrected_BGR = fdr_io::lab2bgr(corrected_Lab);
        corrected_BGR.convertTo(corrected_BGR, CV_8U, 255.0);
    } else {              // the same epilogue in two device passes (fdr_white_balance_u8)
        const float* orig[3]; const float* rest[3];
        vector<Mat> keep_o, keep_r;
        for (int c = 0; c < 3; ++c) {
            keep_o.push_back(input[c].isContinuous() ? input[c] : input[c].clone());
            keep_r.push_back(channels[c].isContinuous() ? channels[c] : channels[c].clone());
        }
        for (int c = 0; c < 3; ++c) { orig[c] = keep_o[c].ptr<float>(0); rest[c] = keep_r[c].ptr<float>(0); }
        corrected_BGR = Mat(img.rows, img.cols, CV_8UC3);
        if (fdr_white_balance_u8(0, orig, rest, img.rows, img.cols, img.cols, corrected_BGR.ptr<unsigned char>(0), 3 * img.cols) != FDR_OK) {
            cerr << "Error: " << __FILE__ << ":" << __LINE__ << ", " << fdr_last_error() << "\n";
            exit(1);
        }
    }
    if (!out_path.empty()) {
        if (!fdr_io::imwrite(out_path, corrected_BGR)) { cout << "Cannot write " << out_path << "\n"; return -1; }
        cout << "Wrote " << out_path << "\n";
    }
    return 0;
}
