/*
 * slam_oracle_main.c -- command-line front end of the CPU restatement, same
 * CLI as the reference binary (src/slam/main.cpp:100-112):
 *     slam_oracle <featurefile> [--dump FILE] < sensors
 * TEST INFRASTRUCTURE ONLY (see ekf_oracle.h).
 */
#include "ekf_oracle.h"
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc < 2 || !strcmp(argv[1], "-h")) {
        fprintf(stderr, "Usage: %s [measurements] < sensors\n", argv[0]);
        exit(EXIT_FAILURE);
    }
    FILE *fp = fopen(argv[1], "r");
    if (!fp) { perror("fopen body"); exit(EXIT_FAILURE); }
    FILE *dump = NULL;
    if (argc >= 4 && !strcmp(argv[2], "--dump")) dump = fopen(argv[3], "w");
    int rc = orc_slam_run(stdin, fp, stdout, stderr, dump, NULL);
    if (dump) fclose(dump);
    fclose(fp);
    return rc;
}
