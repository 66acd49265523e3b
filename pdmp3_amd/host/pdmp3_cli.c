/* pdmp3_cli -- command-line front end of libpdmp3.so.
 *
 * Contract of the reference's driver program (main.c:1-6): every argument is a file name; they are handed to
 * pdmp3() (include/pdmp3.h) without the program name, which decodes them in order and appends interleaved int16
 * PCM to "<first name>.raw" ("-" reads stdin and writes stdout).  Exit status 1 when no file is named, else 0. */
#include <stdio.h>

#include "../../include/pdmp3.h"

int main(int argc, char* argv[]) {
  if (argc <= 1) {
    fputs("usage: pdmp3_cli FILE.mp3 [FILE.mp3 ...]\n"
          "       decodes MPEG-1 Layer III on the GPU; PCM (int16, interleaved) goes to <first FILE>.raw\n", stderr);
    return 1;
  }
  char* const* files = &argv[1];
  pdmp3(files);
  return 0;
}
