#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (run in the authoring
container only: /root/reference does not exist on the GPU box).

What it produces is DATA only:
  * byte copies of the reference's own test fixtures (tests/resources/*.rpw, the
    16 kHz i16 wavs) -- the .rpw files hold the reference's own MFCC output
    (written by tests/wakeword.rs:27-54), i.e. golden vectors for the MFCC path;
  * expectations.json: the exact f32 values asserted by tests/detector.rs, typed
    in by hand below with the line they come from.
No reference source text is copied.
"""
import json
import os
import shutil
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

FILES = [
    "oye_casa_g.rpw", "oye_casa_g_v2.rpw", "alexa.rpw", "ok_casa-tiny.rpw",
    "oye_casa_g_1.wav", "oye_casa_g_2.wav", "oye_casa_g_3.wav", "oye_casa_g_4.wav", "oye_casa_g_5.wav",
    "alexa.wav", "alexa2.wav", "alexa3.wav",
    # 48 kHz f32 recordings: everything that goes through the resampler (src/audio/encoder.rs:72-83)
    "oye_casa_real.rpw", "oye_casa_real_1.wav", "oye_casa_real_2.wav", "oye_casa_real_3.wav", "oye_casa_real_4.wav",
    "oye_casa_real_5.wav", "oye_casa_real_6.wav", "real_sample.wav", "ok_casa.wav",
    # written by src/audio/encoder.rs:139-183 (reencode_wav_with_different_format): oye_casa_g_1.wav through
    # rencode_and_resample::<i16> in 480-sample frames -- pins Sample::into_f32 for i16
    "oye_casa_g_1_f32.wav",
    # written by the filters' own tests from real_sample.wav (48 kHz -> resampler -> filter), f32 16 kHz:
    # src/audio/band_pass_filter.rs:69-122 / :124-185, src/audio/gain_normalizer_filter.rs:81-131
    "band-pass_example.wav", "gain-normalizer_example.wav", "gain_normalized_band-pass_example.wav",
]
# tests/resources/{train,test}: the labelled files are byte copies of wavs listed above (oye_casa_real_{1,3,4,5}.wav with
# "[oye casa]" in the name, test/oye_casa_g_2[oye casa].wav); only the noise recordings are new
TRAIN_FILES = ["train/noise0.wav", "train/noise1.wav", "test/noise3.wav", "test/noise4.wav"]

# tests/detector.rs -- simulation stream: 5 s zeros + oye_casa_g_1.wav[44:] + 5 s zeros +
# oye_casa_g_2.wav[44:] + 5 s zeros, i16 LE bytes, fed in get_bytes_per_frame() chunks (:361-426)
EXPECT = {
    "source": "rustpotter v3.0.2 tests/detector.rs",
    "simulation": {
        "max_v2": {"lines": "9-22", "rpw": "oye_casa_g_v2.rpw", "score_mode": "max", "avg_threshold": 0.2,
                   "threshold": 0.5, "detections": [[0.6495044, 0.7310586], [0.5804737, 0.721843]]},
        "max": {"lines": "24-38", "rpw": "oye_casa_g.rpw", "score_mode": "max", "avg_threshold": 0.2,
                "threshold": 0.5, "detections": [[0.6495044, 0.7310586], [0.5804737, 0.721843]]},
        "median": {"lines": "40-54", "rpw": "oye_casa_g.rpw", "score_mode": "median", "avg_threshold": 0.2,
                   "threshold": 0.5, "detections": [[0.64608675, 0.60123634], [0.5288923, 0.63968724]]},
        "average": {"lines": "56-70", "rpw": "oye_casa_g.rpw", "score_mode": "average", "avg_threshold": 0.2,
                    "threshold": 0.5, "detections": [[0.64608675, 0.60458726], [0.5750509, 0.6313083]]},
        "vad_easy": {"lines": "72-87", "rpw": "oye_casa_g.rpw", "score_mode": "max", "avg_threshold": 0.2,
                     "threshold": 0.5, "vad_mode": "easy",
                     "detections": [[0.6495044, 0.7310586], [0.5804737, 0.721843]]},
        "ignore_alexa": {"lines": "89-99", "rpw": "alexa.rpw", "score_mode": "max", "avg_threshold": 0.0,
                         "threshold": 0.45, "min_scores": 0, "detections": []},
        "ignore_alexa_filters": {"lines": "100-111", "rpw": "alexa.rpw", "score_mode": "max", "avg_threshold": 0.0,
                                 "threshold": 0.45, "min_scores": 0, "gain_normalizer": True, "band_pass": True,
                                 "detections": []},
        "band_pass": {"lines": "112-127", "rpw": "oye_casa_g.rpw", "score_mode": "max", "avg_threshold": 0.0,
                      "threshold": 0.5, "band_pass": True, "low_cutoff": 80.0, "high_cutoff": 400.0,
                      "detections": [[None, 0.6858197], [None, 0.66327363]]},
        "gain_normalizer": {"lines": "129-143", "rpw": "oye_casa_g.rpw", "score_mode": "max", "avg_threshold": 0.0,
                            "threshold": 0.5, "gain_normalizer": True, "gains": [0.2, 5.0],
                            "detections": [[None, 0.7304294], [None, 0.71067876]]},
        "gain_and_band_pass": {"lines": "145-162", "rpw": "oye_casa_g.rpw", "score_mode": "median",
                               "avg_threshold": 0.0, "threshold": 0.5, "gain_normalizer": True, "band_pass": True,
                               "low_cutoff": 80.0, "high_cutoff": 500.0, "gains": [0.2, 5.0],
                               "detections": [[None, 0.5775406], [None, 0.5828697]]},
    },
    # tests/detector.rs:163-213: 48 kHz f32 wav + 5 s of zeros fed in get_samples_per_frame() chunks (:300-325);
    # detections are [avg_score, score, counter]
    "audio_file": {
        "noise": {"lines": "163-186", "rpw": "oye_casa_real.rpw", "wav": "real_sample.wav", "score_mode": "max",
                  "avg_threshold": 0.3, "threshold": 0.47, "min_scores": 5,
                  "detections": [[0.4676845, 0.527971, 24], [0.32865646, 0.48120698, 7], [0.30807483, 0.5164661, 35]]},
        "noise_filters": {"lines": "188-213", "rpw": "oye_casa_real.rpw", "wav": "real_sample.wav", "score_mode": "max",
                          "avg_threshold": 0.3, "threshold": 0.49, "min_scores": 5, "gain_normalizer": True,
                          "min_gain": 0.4, "band_pass": True, "low_cutoff": 210.0, "high_cutoff": 700.0,
                          "detections": [[0.45496628, 0.5380342, 23], [0.336222, 0.5001262, 5], [0.3049497, 0.5189481, 31]]},
    },
    # the filter example wavs: resampler output of real_sample.wav through the front-end, sample by sample
    "filter_examples": {
        "band-pass_example.wav": {"lines": "band_pass_filter.rs:69-122", "band_pass": True, "low_cutoff": 80.0, "high_cutoff": 400.0},
        "gain-normalizer_example.wav": {"lines": "gain_normalizer_filter.rs:81-131", "gain_normalizer": True, "gain_ref": 0.003,
                                        "min_gain": 0.1, "max_gain": 1.0},
        "gain_normalized_band-pass_example.wav": {"lines": "band_pass_filter.rs:124-185", "gain_normalizer": True, "gain_ref": 0.003,
                                                  "min_gain": 0.1, "max_gain": 1.0, "band_pass": True, "low_cutoff": 80.0,
                                                  "high_cutoff": 400.0},
    },
    # tests/detector.rs:216-267: the wakeword-model runs on ok_casa.wav (48 kHz): [counter, avg_score, score, label logit, none logit]
    "audio_file_nn": {
        "model": {"lines": "216-232", "avg_threshold": 0.0, "detections": [[34, 0.0, 0.9997649, 3.7506533, -16.83091]]},
        "model_avg": {"lines": "234-250", "avg_threshold": 0.5, "detections": [[34, 0.9997649, 0.9997649, 3.7506533, -16.83091]]},
        "model_eager": {"lines": "252-267", "avg_threshold": 0.0, "min_scores": 20, "eager": True,
                        "detections": [[20, 0.0, 0.9992142, 23.990948, 6.0654087]]},
    },
    # tests/detector.rs:216-232: score = calc_inverse_similarity(label, none, score_ref*10)
    "nn_score_formula": {"lines": "216-232", "label_logit": 3.7506533, "none_logit": -16.83091,
                         "score_ref": 0.22, "score": 0.9997649},
    # tests/detector.rs:252-267 (eager): logits 23.990948 / 6.0654087 -> 0.9992142
    "nn_score_formula_eager": {"lines": "252-267", "label_logit": 23.990948, "none_logit": 6.0654087,
                               "score_ref": 0.22, "score": 0.9992142},
    # tests/wakeword.rs:86-98: ModelType::Medium, lr 0.027, 10 epochs, mfcc_size 16 on tests/resources/{train,test}
    "train": {"lines": "86-98", "m_type": "medium", "learning_rate": 0.027, "epochs": 10, "test_epochs": 10, "mfcc_size": 16,
              "train": {"noise0.wav": "noise0.wav", "noise1.wav": "noise1.wav",
                        "oye_casa_real_1[oye casa].wav": "oye_casa_real_1.wav", "oye_casa_real_3[oye casa].wav": "oye_casa_real_3.wav",
                        "oye_casa_real_4[oye casa].wav": "oye_casa_real_4.wav", "oye_casa_real_5[oye casa].wav": "oye_casa_real_5.wav"},
              "test": {"noise3.wav": "noise3.wav", "noise4.wav": "noise4.wav", "oye_casa_g_2[oye casa].wav": "oye_casa_g_2.wav"},
              "labels": 2, "weights": 6, "train_size": 168},
    # frame counts of the reference's own MFCC output, SURVEY.md §4
    "rpw_shapes": {"oye_casa_g.rpw": {"oye_casa_g_1.wav": 108, "oye_casa_g_2.wav": 96, "oye_casa_g_3.wav": 90,
                                      "oye_casa_g_4.wav": 93, "oye_casa_g_5.wav": 102},
                   "alexa.rpw": {"alexa.wav": 117, "alexa2.wav": 126, "alexa3.wav": 99}},
}

if __name__ == "__main__":
    for f in FILES:
        shutil.copyfile(os.path.join(REF, "tests", "resources", f), os.path.join(HERE, f))
    for f in TRAIN_FILES:
        shutil.copyfile(os.path.join(REF, "tests", "resources", f), os.path.join(HERE, os.path.basename(f)))
    with open(os.path.join(HERE, "expectations.json"), "w") as fh:
        json.dump(EXPECT, fh, indent=1)
    print("wrote", len(FILES), "fixture files + expectations.json")
